#!/usr/bin/env python3
"""Headline benchmark: proofs verified / second (BN254, k = 14) on N MI355X of one node.

One "step" = one pass of the whole hot path (point decompression -> Blake2b transcript -> Fr program -> shared-base fold
-> pooled MSMs -> one pairing) over one batch of `--batch` proofs PER GPU that is already resident in HBM.  N > 1: every
rank verifies its own shard, one 1312-byte accumulator record per rank and step (the two accumulators in pieces + the shard's failed-proof
count) is all-gathered over RCCL and folded, and a single pairing closes the whole N x batch step (weak scaling).
The tail of a step — window Horner, the pairing — is a handful of waves with long dependent chains, so steps are issued
`--groups` at a time as one grouped batch (h2v_batch_set_groups: every kernel runs once for all of them, each step keeps
its own accumulators and its own pairing) and `--depth` such launches are in flight (one HIP stream each); all K timed
steps start and finish inside the timed region.

`python bench.py --gpus N` started as a plain process launches the N ranks itself (halo2_verifier_amd/launch.py: N fresh
child processes, the parent never touches the GPU); under torch.distributed.run the ranks come from the environment.  Either
way WORLD_SIZE must equal --gpus, otherwise the run fails instead of silently measuring another job.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel of the MSM (msm_accumulate): achieved = 96 B x the terms of a
launch / the kernel's average duration inside the timed region (its dispatch's own timestamps); `roofline.stage` has the whole MSM stage.
`cpu_baseline` = the CPU oracle (a port of the reference algorithm, single thread like the reference) timed on a bounded
sample of the same proofs on this host.
"""
import argparse
import ctypes
import datetime
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

K_CIRCUIT = 14
N_PUBLIC = 8
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MAX_HW_QUEUES = 16      # see hw_queue_env()
SHORT_RUN_S = 0.05      # a timed region shorter than this is repeated and its median reported (--repeats)


def hw_queue_env():
    """A batch's tail (window Horner, pairing) is one or two waves; throughput comes from several launches in flight, each on
    its own HIP stream.  The ROCm runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), which would
    serialise most of them; ask for 16 before the runtime initialises.  More is not better (profiles/r02_hw_queues.txt: 24, 32 and
    64 queues all run to completion at the same ~10 M proofs/s; round 1's "aborts at 32" was not reproducible), so an inherited
    larger value is clamped to the measured setting rather than trusted."""
    try:
        v = int(os.environ.get("GPU_MAX_HW_QUEUES", MAX_HW_QUEUES))
    except ValueError:
        v = MAX_HW_QUEUES
    if os.environ.get("H2V_BENCH_RAW_HW_QUEUES") == "1":   # diagnosis only: take the inherited value as it is
        return
    os.environ["GPU_MAX_HW_QUEUES"] = str(max(1, min(v, MAX_HW_QUEUES)))


def load_or_make_proofs(count, k, log):
    """`count` distinct proofs of the vector_mul circuit, cached on disk."""
    import circuits
    cache = os.path.join(ROOT, ".bench_cache")
    os.makedirs(cache, exist_ok=True)
    tag = f"vm_k{k}_pub{N_PUBLIC}_n{count}"
    paths = {x: os.path.join(cache, f"{tag}.{x}") for x in ("proofs", "inst", "vk", "params")}
    if all(os.path.exists(p) for p in paths.values()):
        d = {x: open(p, "rb").read() for x, p in paths.items()}
        if len(d["proofs"]) == count * 1024 and len(d["inst"]) == count * 32 * N_PUBLIC:
            return d
    t = time.time()
    s = circuits.setup_vector_mul(k, N_PUBLIC, s_seed=0x48325630)  # known-s test SRS, s from seed "H2V0"
    threads = min(32, os.cpu_count() or 1)
    P, I = circuits.prove_vector_mul_batch(s, count, seed=0x48325630, threads=threads)
    d = dict(proofs=b"".join(P), inst=b"".join(b"".join(col) for inst in I for col in inst), vk=s.vk, params=s.params)
    s.free()
    for x, p in paths.items():
        with open(p + ".tmp", "wb") as f:
            f.write(d[x])
        os.replace(p + ".tmp", p)
    log(f"generated {count} proofs (k={k}) with {threads} threads in {time.time() - t:.1f}s")
    return d


def load_or_make_wide(count, k, log):
    """BASELINE.json config 4: `count` distinct proofs of the lookup-heavy VK (32 advice, 16 fixed, 8 two-column lookups, gate degree
    5: 74 points + 170 scalars = 7808-byte proofs, 124 right-channel terms), cached on disk.  k = 10 here: the test prover needs
    ~70 s per proof at the k = 16 BASELINE.json names (that size is covered for parity by tests/golden/wide_k16_lookup_heavy.json);
    the verifier's work per proof does not depend on k beyond the k squarings of x^n."""
    import concurrent.futures
    import circuits
    cache = os.path.join(ROOT, ".bench_cache")
    os.makedirs(cache, exist_ok=True)
    tag = f"wide_k{k}_a32_f16_l8_d5_n{count}"
    paths = {x: os.path.join(cache, f"{tag}.{x}") for x in ("proofs", "inst", "vk", "params")}
    if all(os.path.exists(p) for p in paths.values()):
        d = {x: open(p, "rb").read() for x, p in paths.items()}
        if len(d["proofs"]) % count == 0 and len(d["inst"]) == count * 32 * 8:
            return d
    t = time.time()
    s = circuits.setup_wide(k, A=32, F=16, L_=8, Sh=0, deg=5)
    threads = min(32, os.cpu_count() or 1)
    with concurrent.futures.ThreadPoolExecutor(max_workers=threads) as ex:   # ctypes releases the GIL; the setup is read-only
        pairs = list(ex.map(lambda i: circuits.prove_wide(s, witness_seed=i, rng_seed=1000 + i), range(count)))
    d = dict(proofs=b"".join(p for p, _ in pairs), inst=b"".join(b"".join(col) for _, inst in pairs for col in inst), vk=s.vk, params=s.params)
    s.free()
    for x, p in paths.items():
        with open(p + ".tmp", "wb") as f:
            f.write(d[x])
        os.replace(p + ".tmp", p)
    log(f"generated {count} lookup-heavy proofs (k={k}) with {threads} threads in {time.time() - t:.1f}s")
    return d


def oracle_group(d, n, draws):
    """(ok, left_xy, right_xy) of the CPU oracle for the first n bench proofs under the given draws (n x 32 bytes): the parity
    reference for one timed group of the benchmark."""
    import oracle_lib
    L = oracle_lib.load()
    cl = (ctypes.c_size_t * 1)(N_PUBLIC)
    st = (ctypes.c_int * n)()
    ok = ctypes.c_int(0)
    left, right = ctypes.create_string_buffer(64), ctypes.create_string_buffer(64)
    rc = L.h2o_verify_batch(d["params"], len(d["params"]), 1, d["vk"], len(d["vk"]), 1, n, d["proofs"][: n * 1024], 1024, d["inst"][: n * 32 * N_PUBLIC], cl, 1, draws, st,
                            ctypes.byref(ok), left, right)
    assert rc == 0
    return bool(ok.value), left.raw, right.raw


def cpu_baseline(d, sample, log):
    """CPU oracle (oracle/: a C++ port that follows the reference algorithm step for step) on a bounded sample of the bench
    proofs, on this host.  Headline figures: AccumulatorStrategy semantics (N x verify_proof + one pairing, including the
    reference's O(N^2) re-scaling), ONE thread — the reference is single-threaded (arithmetic.rs:127-134).  Also reported:
    the same with one proof-shard per host core (each shard its own accumulator and pairing), and SingleStrategy
    (one pairing per proof) on one thread."""
    import concurrent.futures
    import oracle_lib
    L = oracle_lib.load()
    cl = (ctypes.c_size_t * 1)(N_PUBLIC)

    def accumulate(first, n):
        rand = b"".join((i * 0x9e3779b97f4a7c15 + 12345).to_bytes(32, "little") for i in range(first + 1, first + n + 1))
        st = (ctypes.c_int * n)()
        ok = ctypes.c_int(0)
        left, right = ctypes.create_string_buffer(64), ctypes.create_string_buffer(64)
        rc = L.h2o_verify_batch(d["params"], len(d["params"]), 1, d["vk"], len(d["vk"]), 1, n, d["proofs"][first * 1024:(first + n) * 1024], 1024,
                                d["inst"][first * 32 * N_PUBLIC:(first + n) * 32 * N_PUBLIC], cl, 1, rand, st, ctypes.byref(ok), left, right)
        assert rc == 0 and ok.value == 1
        return n

    n = sample
    t0 = time.perf_counter()
    accumulate(0, n)
    dt = time.perf_counter() - t0
    out = dict(value=n / dt, unit="proofs/s", cores=1, kind="port",
               sample=f"{n} of the bench proofs, AccumulatorStrategy (N x verify_proof + 1 pairing), oracle/ C++ port, 1 thread, {dt:.2f}s")
    # SingleStrategy, one thread
    m = min(n, 512)
    st = (ctypes.c_int * m)()
    t0 = time.perf_counter()
    acc = L.h2o_verify_each(d["params"], len(d["params"]), 1, d["vk"], len(d["vk"]), 1, m, d["proofs"][: m * 1024], 1024, d["inst"][: m * 32 * N_PUBLIC], cl, 1, st)
    dt1 = time.perf_counter() - t0
    assert acc == m
    out["single_strategy"] = dict(value=m / dt1, unit="proofs/s", cores=1, sample=f"{m} proofs, one pairing per proof, {dt1:.2f}s")
    # all host cores: one shard (own accumulator + pairing) per thread; ctypes releases the GIL around the call
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    per = max(64, n // cores)
    while per * cores > n and cores > 1:   # the sample holds n proofs
        cores -= 1
    t0 = time.perf_counter()
    with concurrent.futures.ThreadPoolExecutor(max_workers=cores) as ex:
        done = sum(ex.map(lambda t: accumulate(t * per, per), range(cores)))
    dtn = time.perf_counter() - t0
    out["all_cores"] = dict(value=done / dtn, unit="proofs/s", cores=cores, sample=f"{done} proofs, {cores} threads x {per} (one accumulator + pairing per thread), {dtn:.2f}s")
    return out


def launch_shape(steps, groups, depth):
    """How K steps are cut into launches.  groups / depth = 0 -> automatic: a launch's latency chain (transcript -> Fr program ->
    MSM tail -> pairing) is several milliseconds whatever it carries, so a short run is spread over several launches in flight
    (their chains overlap) and a long one fills every launch (fewer, larger kernels)."""
    if depth <= 0:
        depth = 8
    if groups <= 0:
        # measured (gpurun_out/r02_p1): for a short run one launch that carries every step beats several smaller launches in
        # flight — their latency chains overlap, but every launch pays ~30 kernel submissions on the host
        groups = 32
    groups = max(1, min(groups, steps))
    launches, rem = steps // groups, steps % groups
    depth = max(1, min(depth, launches))
    return groups, launches, rem, depth


def traffic_record(terms_per_launch):
    """HBM traffic of the MSM stage per launch from the committed PMC passes (separate rocprofv3 --pmc runs cannot be taken
    inside the timed region).  The record of this launch shape if there is one, otherwise the nearest shape scaled per term."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_msm_traffic*.json"))):
        try:
            tj = json.load(open(path))
            t, b = tj["terms_per_launch"], tj["msm_stage_traffic_bytes_per_launch"]
        except Exception:
            continue
        if not t or not b:
            continue
        key = (abs(t - terms_per_launch), [-ord(c) for c in os.path.basename(path)])   # same shape: the latest round's file (names sort by round; mtimes do not survive a checkout)
        if best is None or key < best[0]:
            best = (key, path, t, b, tj)
    if best is None:
        return None, "no PMC profile committed", None
    _, path, t, b, tj = best
    rel = os.path.relpath(path, ROOT)
    if t == terms_per_launch:
        return b, f"{rel} (same launch shape: {t} terms per launch; fetch_factor x FETCH_SIZE + WRITE_SIZE with the per-kernel factor of profiles/r03_fetch_calibration.txt: 1 for msm_accumulate's gathers, 2 for streams and whole-line reads)", tj
    return b * terms_per_launch / t, f"{rel} scaled per term ({t} -> {terms_per_launch} terms per launch; per-kernel fetch factors of profiles/r03_fetch_calibration.txt)", tj


def _median(v):
    v = sorted(v)
    return v[len(v) // 2]


def config4_leg(h2v, args, log):
    """BASELINE.json config 4 on one GPU: batches of 1024 lookup-heavy proofs, 8 batches per launch, 4 launches in flight, every
    batch its own accumulators and pairing — the same pipeline as the headline on the VK that stresses expression evaluation."""
    import torch
    k, B, G, depth, launches = 10, 1024, 8, 4, 12
    d = load_or_make_wide(B, k, log)
    plen = len(d["proofs"]) // B
    ctx = h2v.Context(h2v.ParamsKZG(d["params"], h2v.SerdeFormat.RawBytes), h2v.VerifyingKey(d["vk"], h2v.SerdeFormat.RawBytes))
    shape = ctx.proof_shape()
    if args.tuning:   # measurement aid: forced kernel variants (h2v_ctx_set_tuning), e.g. --tuning msm_acc_waves=4
        ctx.set_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in args.tuning.split(","))})
    tail = b"".join(((i * 0x9e3779b97f4a7c15 + 77) % (1 << 250)).to_bytes(32, "little") for i in range(1, B * G + 1))
    bs = []
    for _ in range(depth):
        b = h2v.Batch(ctx, B * G, 8, groups=G)
        b.upload(d["proofs"] * G, plen, d["inst"] * G, [8], tail)
        bs.append(b)
    fl = [False] * depth

    def retire(i):
        ok, st, _, _ = bs[i].finish_groups(raw_statuses=True)
        fl[i] = False
        if not all(ok) or st.count(0) != len(st):
            raise SystemExit("verification failed inside the config-4 leg")

    def run(n):
        for k2 in range(n):
            i = k2 % depth
            if fl[i]:
                retire(i)
            bs[i].launch(True); fl[i] = True
        for i in range(depth):
            if fl[i]:
                retire(i)
    run(depth)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); run(launches); dt = time.perf_counter() - t0
    # one batch alone (latency)
    b1 = h2v.Batch(ctx, B, 8)
    b1.upload(d["proofs"], plen, d["inst"], [8], tail[: 32 * B]); b1.set_profiling(True)
    alone = []
    for _ in range(5):
        t1 = time.perf_counter(); b1.launch(True); ok1, st1, _, _ = b1.finish(); alone.append(time.perf_counter() - t1)
        assert ok1
    stages = b1.timings_ms()
    b1.close()
    for b in bs:
        b.close()
    ctx.close()
    return {"workload": f"lookup-heavy VK (32 advice, 16 fixed, 8 two-column lookups, gate degree 5), k={k} (BASELINE.json names k=16: parity at that size by tests/golden/wide_k16_lookup_heavy.json; "
                        f"the test prover needs ~70 s per k=16 proof), {B} distinct proofs of {plen} B ({shape['n_points']} points, {shape['n_scalars']} scalars, {shape['n_right_terms']} right-channel terms), "
                        f"SHPLONK/Blake2b, batches of {B}, {G} batches per launch, {depth} launches in flight, {launches} launches timed",
            "value": B * G * launches / dt, "unit": "proofs/s", "ms_per_batch": dt / (G * launches) * 1e3,
            "one_batch_alone_ms": _median(alone) * 1e3, "one_batch_alone_stages_ms": stages}


def single_strategy_leg(h2v, ctx, d, args):
    """SingleStrategy on the GPU (one pairing per proof, kzg/strategy.rs:143-181): 512 one-proof groups per launch — what
    h2v_verify_each runs — on the bench proofs, resident in HBM like the headline; and the one-shot call including packing and upload."""
    n = 512
    ones = b"".join((1).to_bytes(32, "little") for _ in range(n))
    b = h2v.Batch(ctx, n, N_PUBLIC, groups=n)
    b.upload(d["proofs"][: n * 1024], 1024, d["inst"][: n * 32 * N_PUBLIC], [N_PUBLIC], ones)
    times = []
    for i in range(9):
        t0 = time.perf_counter()
        b.launch(True)
        ok, st, _, _ = b.finish_groups(raw_statuses=True)
        times.append(time.perf_counter() - t0)
        if not all(ok) or st.count(0) != len(st):
            raise SystemExit("verification failed inside the SingleStrategy leg")
    b.close()
    P = [d["proofs"][i * 1024:(i + 1) * 1024] for i in range(n)]
    I = [[[d["inst"][(i * N_PUBLIC + j) * 32:(i * N_PUBLIC + j + 1) * 32] for j in range(N_PUBLIC)]] for i in range(n)]
    ctx.verify_each(P[:8], I[:8])
    t0 = time.perf_counter(); st = ctx.verify_each(P, I); t_api = time.perf_counter() - t0
    assert st == [0] * n
    ones_t = []
    for _ in range(9):   # (the first call after the 512-proof one re-sizes the context's scratch batch: a median, not one sample)
        t0 = time.perf_counter(); st1 = ctx.verify_each(P[:1], I[:1]); ones_t.append(time.perf_counter() - t0)
        assert st1 == [0]
    t_one = _median(ones_t)
    dt = _median(times[2:])
    return {"value": n / dt, "unit": "proofs/s", "sample": f"{n} proofs per launch as one-proof groups (own MSMs, own pairing each), resident in HBM, median of {len(times) - 2} launches, {dt * 1e3:.2f} ms per launch",
            "one_shot_api": {"value": n / t_api, "unit": "proofs/s", "note": f"h2v_verify_each({n} proofs) from host byte strings: packing + upload + launch + results, {t_api * 1e3:.2f} ms"},
            "one_proof_latency_ms": t_one * 1e3,
            "one_proof_latency_note": "h2v_verify_each on ONE proof from host byte strings (packing, upload, launch, own pairing, results), median of 9 calls"}


def rank_main(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    log = (lambda m: print(f"[bench] {m}", file=sys.stderr, flush=True)) if rank == 0 else (lambda m: None)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to measure a different job than the one asked for "
                         f"(start it as `python bench.py --gpus {args.gpus}` or under torch.distributed.run with --nproc-per-node {args.gpus})")
    one_device = os.environ.get("H2V_BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("H2V_BENCH_BACKEND", "nccl")

    if args.dry_run:
        # the launch plan, no GPU: what the N ranks would do, plus a rendezvous + collective over gloo to prove the job is N ranks
        import torch
        import torch.distributed as dist
        from halo2_verifier_amd import distributed as h2d
        G, launches, rem, depth = launch_shape(args.steps, args.groups, args.depth)
        total = args.batch * world
        lo, hi = h2d.shard_bounds(total, world, rank)
        seen = world
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
            t = torch.ones(1, dtype=torch.int64)
            dist.all_reduce(t)
            seen = int(t.item())
            dist.barrier()
            dist.destroy_process_group()
        print(f"[bench] dry run: rank {rank}/{world} local_rank {local_rank} shard [{lo}, {hi}) of {total} proofs per step", file=sys.stderr, flush=True)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_seen": seen, "steps": args.steps, "warmup": args.warmup, "steps_per_launch": G,
                              "launches": launches, "remainder_steps": rem, "pipeline_depth": depth, "proofs_per_gpu_per_step": args.batch}), flush=True)
        return

    hw_queue_env()
    import torch
    import torch.distributed as dist
    import halo2_verifier_amd as h2v
    from halo2_verifier_amd import distributed as h2d

    need = 1 if one_device else (local_rank + 1 if world == 1 else world)
    if not torch.cuda.is_available() or h2v.device_count() < need:
        raise SystemExit(f"bench.py needs {need} HIP device(s) for --gpus {args.gpus} (found {h2v.device_count()}): the product path has no CPU fallback")
    # rehearsal knobs (not used by the driver): H2V_BENCH_BACKEND=gloo and H2V_BENCH_ONE_DEVICE=1 run an N-rank job on a
    # single-GPU box (all ranks on cuda:0, collectives staged through the host) to exercise the multi-rank control flow
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=600))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))

    def barrier():
        if world > 1:
            if backend == "nccl":
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()

    # inputs: rank 0 generates (or loads) the proofs, the others wait and read the cache
    if rank == 0:
        d = load_or_make_proofs(args.distinct, K_CIRCUIT, log)
    barrier()
    if rank != 0:
        d = load_or_make_proofs(args.distinct, K_CIRCUIT, log)

    ctx = h2v.Context(h2v.ParamsKZG(d["params"], h2v.SerdeFormat.RawBytes), h2v.VerifyingKey(d["vk"], h2v.SerdeFormat.RawBytes), device=local_rank)
    shape = ctx.proof_shape()
    tuning = {k: int(v) for k, v in (kv.split("=") for kv in args.tuning.split(","))} if args.tuning else {}
    if tuning:   # measurement aid: forced kernel variants (h2v_ctx_set_tuning), e.g. --tuning msm_acc_waves=4; the line says so ("tuning")
        ctx.set_tuning(**tuning)

    def measure(B, steps, warmup, groups, depth_arg, reupload=False, isolated_launches=0, repeats=1):
        """Time exactly `steps` steps of B proofs per GPU.  A step is one batch of B proofs per GPU with its own accumulators
        and its own pairing (AccumulatorStrategy).  One launch carries G steps side by side (a grouped batch); `depth` launches
        are in flight; a remainder launch completes --steps.  Returns a dict of timings."""
        G, launches, rem, depth = launch_shape(steps, groups, depth_arg)
        warm_launches = (warmup + G - 1) // G
        reps = (B + args.distinct - 1) // args.distinct
        proofs_one = (d["proofs"] * reps)[: B * 1024]
        inst_one = (d["inst"] * reps)[: B * 32 * N_PUBLIC]
        total = B * world
        lo, hi = h2d.shard_bounds(total, world, rank)
        # per group one seeded stream of Fr::random draws for the whole (N x batch) step, indexed by global proof id; a rank
        # uploads, for every group, the draws from its first proof to the end of the step (the multiplier of a proof is the
        # product of the draws of all later proofs of its step)
        tails = []
        for g in range(G):
            rand_all = b"".join(((i * 0x9e3779b97f4a7c15 + 0x1234567 + g * 0x51ed27) % (1 << 250)).to_bytes(32, "little") for i in range(1, total + 1))
            tails.append(h2d.tail_for_shard(rand_all, lo))
        # objects 0 .. depth-1 carry G steps each; object `depth` (if any) carries the `rem` steps that complete --steps
        sizes = [G] * depth + ([rem] if rem else [])
        streams = [torch.cuda.Stream(device=local_rank) for _ in sizes]
        # the library's sharded-batch object (halo2_verifier_amd/distributed.py): at N = 1 a launch ends in its own pairings, at
        # N > 1 it runs shard -> export -> RCCL all-gather of G x 1312 B per rank -> fold -> ONE pairing per step
        shards, batches = [], []
        for s, g in zip(streams, sizes):
            sb = h2d.ShardedBatch(ctx, B * g, N_PUBLIC, groups=g, stream=s, device=dev,
                                  always_exchange=bool(os.environ.get("H2V_BENCH_FORCE_SHARDED")))   # (the knob runs the N > 1 call sequence on one GPU: what sharding costs besides the collective)
            sb.upload(proofs_one * g, 1024, inst_one * g, [N_PUBLIC], b"".join(tails[:g]))   # resident in HBM before the timed region
            sb.batch.set_profiling(h2v.Batch.PROFILE_KERNEL)   # timed region: only the dominant kernel's own timestamps (stage events are barrier packets)
            shards.append(sb); batches.append(sb.batch)
        in_flight = [False] * len(sizes)
        last_group0 = [None]
        stage_sum = {k: 0.0 for k in h2v.Batch.STAGES}
        stage_cnt = [0]
        uploads = [(proofs_one * g, inst_one * g, b"".join(tails[:g])) for g in sizes] if reupload else None

        def retire(i, timed):
            ok, st, left, right = shards[i].finish(raw_statuses=True)   # every proof's status, as the C array's bytes: all zero <=> all OK
            in_flight[i] = False
            if not all(ok) or st.count(0) != len(st):
                raise SystemExit(f"verification failed inside the benchmark: ok={ok}")
            if timed and i == 0:
                last_group0[0] = (left[0], right[0])   # accumulators of timed group 0: compared with the CPU oracle after the timed region
            if timed and i < depth:   # stage statistics are per full launch
                for k2, v in batches[i].timings_ms().items():
                    stage_sum[k2] += v
                stage_cnt[0] += 1

        def submit(i):
            if uploads:   # PCIe-inclusive leg: host buffers -> device again, the copy chunked under the decompression (h2v_batch_upload_launch)
                shards[i].upload_launch(uploads[i][0], 1024, uploads[i][1], [N_PUBLIC], uploads[i][2])
            else:
                shards[i].launch()
            in_flight[i] = True

        def run(n_launches, timed, with_rem):
            for step in range(n_launches):
                i = step % depth
                if in_flight[i]:
                    retire(i, timed)
                submit(i)
            if rem and with_rem:
                submit(depth)
            for i in range(len(sizes)):
                if in_flight[i]:
                    retire(i, timed)

        def timed_region():
            """EXACTLY `steps` steps between barrier + synchronize on both sides; the maximum over ranks."""
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(launches, True, True)
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            dt1 = time.perf_counter() - t0
            if world > 1:
                tmax = torch.tensor([dt1], dtype=torch.float64, device=dev)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                dt1 = float(tmax.item())
            return dt1

        run(warm_launches, False, True)
        runs = [timed_region()]
        # A timed region of a few milliseconds is ONE sample of a latency chain (the driver's `--steps 20` is a single 3 ms launch;
        # between boxes of the pool it moved by +-4 %): such a region is repeated — the same K steps, the same launches — and the
        # MEDIAN is what the line reports, with every run listed.  Every rank takes the same decision (the times are the max over ranks).
        if repeats > 1 and runs[0] < SHORT_RUN_S:
            runs += [timed_region() for _ in range(repeats - 1)]
        dt = sorted(runs)[len(runs) // 2]
        stages = {k2: v / max(stage_cnt[0], 1) for k2, v in stage_sum.items()}
        # Outside the timed region: a few launches with ONE launch in flight, so that the HIP-event stage times are those of
        # the kernels alone (with several launches in flight a stage's elapsed time includes other streams' kernels).  The
        # roofline figure uses these undisturbed durations; they agree with the rocprofv3 depth-1 summary under profiles/.
        isolated = None
        if isolated_launches:
            batches[0].set_profiling(True)
            for k2 in stage_sum:
                stage_sum[k2] = 0.0
            stage_cnt[0] = 0
            for _ in range(isolated_launches):
                submit(0)
                retire(0, True)
            isolated = {k2: v / max(stage_cnt[0], 1) for k2, v in stage_sum.items()}
        for sb in shards:
            sb.close()
        n_local = hi - lo
        n_shared = max(shape["n_right_terms"] - shape["n_points"], 0)
        # terms of one launch: every point slot of every local proof (right channel) + the VK-wide bases folded
        # over the batch (fixed + permutation commitments + g) + one h2 term per proof (left channel)
        terms_launch = G * (n_local * shape["n_points"] + n_shared + n_local)
        return dict(dt=dt, runs=runs, G=G, launches=launches, rem=rem, depth=depth, total=total, stages=stages, isolated=isolated, terms_launch=terms_launch, B=B,
                    group0=last_group0[0], group0_draws=tails[0] if world == 1 else None)

    m = measure(args.batch, args.steps, args.warmup, args.groups, args.depth, reupload=args.reupload, isolated_launches=4 if world == 1 else 2, repeats=args.repeats)
    m_re = None
    if world == 1 and not args.reupload and not args.no_reupload_leg:
        # PCIe-inclusive leg (SURVEY.md §8d timing method): the same K steps with the host buffers copied to the device again
        # before every launch; reported as value_reupload beside `value`, never instead of it
        m_re = measure(args.batch, args.steps, args.warmup, args.groups, args.depth, reupload=True, repeats=args.repeats)
    m_c3 = None
    if world > 1 and not args.no_config3:
        # BASELINE.json configs 3 and 5: 8192 proofs per GPU per step (65 536 over 8 GPUs), ONE pairing for the whole N x 8192
        # step.  Measured after the headline so that `value` keeps the per-GPU shape of the N = 1 run (weak scaling).
        m_c3 = measure(8192, args.config3_steps, 2, 2, 2)

    if rank == 0:
        G, depth, B = m["G"], m["depth"], m["B"]
        stages, isolated = m["stages"], m["isolated"]
        terms_total = m["terms_launch"]
        src = isolated if isolated else stages
        msm_ms = src["msm"]
        traffic, traffic_src, tj = traffic_record(terms_total)
        achieved = (96.0 * terms_total) / (msm_ms * 1e-3) / 1e9 if msm_ms > 0 else 0.0
        acc_ms = src.get("msm_accumulate", 0.0)
        valu = None
        try:
            va = (tj or {}).get("valu_active", {})
            valu = next((v for k2, v in va.items() if k2 == "h2v::msm_accumulate" or k2.startswith("h2v::msm_accumulate<")), None)
        except Exception:
            pass
        acc_timed = stages.get("msm_accumulate", 0.0) or acc_ms          # the dispatch's own timestamps, every launch of the timed region
        # HBM traffic of the dominant kernel alone, from the same committed PMC record (scaled per term when the shape differs)
        acc_traffic = None
        try:
            pk = (tj or {}).get("per_kernel_bytes", {})
            kb = next((v for k2, v in pk.items() if k2 == "h2v::msm_accumulate" or k2.startswith("h2v::msm_accumulate<")), None)
            if kb is not None and tj.get("terms_per_launch"):
                acc_traffic = kb * terms_total / tj["terms_per_launch"]
        except Exception:
            pass
        acc_gbps = (96.0 * terms_total) / (acc_timed * 1e-3) / 1e9 if acc_timed > 0 else 0.0
        kernels = {"msm_accumulate": {"ms": acc_ms, "ms_timed_region": stages.get("msm_accumulate", 0.0),
                                      "timing": "the dispatch's own start / stop timestamps (hipExtLaunchKernelGGL events), every launch of the timed region; `ms` = the same for launches re-timed one at a time", "alg_GBps": (96.0 * terms_total) / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else None,
                                      "valu_active": valu, "valu_active_source": "SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES from the committed --pmc pass named in traffic_source",
                                      "alu_frac": (2 * 12 * 11 * terms_total) / (acc_ms * 1e-3) / 168e9 if acc_ms > 0 else None,
                                      "alu_frac_note": "Fq products/s of the kernel (11 per mixed addition, 2 x 12 additions per term) over the measured chip-wide peak of the Montgomery product, 168 G/s (tools/limb29_microbench.hip)"}}
        workload = (f"batch {B} proofs/GPU/step, k={K_CIRCUIT}, vector_mul VK (3 advice, 1 fixed, 1 instance col with {N_PUBLIC} public inputs, 4 permutation cols), "
                    f"SHPLONK/Blake2b, AccumulatorStrategy (one pairing per step" + (f" for all {world} x {B} proofs" if world > 1 else "") + f"), {args.distinct} distinct proofs; "
                    f"{G} steps per launch (grouped batch), {depth} launches in flight" + (f", + one launch of {m['rem']} steps" if m["rem"] else ""))
        out = {
            "metric": "proofs verified/sec (BN254, k=14)",
            "value": m["total"] * args.steps / m["dt"],
            "unit": "proofs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": m["dt"] / args.steps * 1e3,
            "ms_per_step_runs": [r / args.steps * 1e3 for r in m["runs"]],
            "timing_note": (f"the timed region ({args.steps} steps) is shorter than {SHORT_RUN_S * 1e3:.0f} ms, so it was run {len(m['runs'])} times (same steps, same launches, each between "
                            "barrier + synchronize); value and ms_per_step are the MEDIAN run, ms_per_step_runs lists all of them in order") if len(m["runs"]) > 1
                           else "one timed region of exactly --steps steps",
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 limbs (254-bit prime-field integers)",
            "data": "synthetic",
            "config": {"workload": workload,
                       "inputs": "copied host -> device before every launch (PCIe-inclusive)" if args.reupload else "resident in HBM before the timed region",
                       "proofs_per_gpu_per_step": B, "steps_per_launch": G, "pipeline_depth": depth, "proof_bytes": shape["proof_len"]},
            # the DOMINANT KERNEL (msm_accumulate: the bucket accumulation of every MSM of a launch): algorithmic bytes of a launch over the
            # kernel's average duration inside the timed region; the whole MSM stage (all its kernels) is in `stage`
            "roofline": {"bound": "hbm", "achieved": acc_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": acc_gbps / HBM_PEAK_GBPS, "traffic": acc_traffic,
                         "traffic_source": traffic_src,
                         "kernel": "msm_accumulate (the dominant kernel of the MSM stage: one mixed G1 addition per entry of the sorted (term, half, window) list of every MSM of a launch)",
                         "kernel_ms": acc_timed, "kernel_ms_one_launch_in_flight": acc_ms,
                         "timing": "the dispatch's own start / stop timestamps on the stream it is launched on (hipExtLaunchKernelGGL events), averaged over every launch of the timed region",
                         "terms_per_launch": terms_total, "algorithmic_bytes_per_launch": 96 * terms_total,
                         "kernels": kernels,
                         "stage": {"what": "the whole MSM stage (msm_glv_prep, msm_sort_lds, msm_seg_scan, msm_accumulate, msm_accumulate_redo, msm_fixup, msm_window, msm_final_parts; both channels of every step of a launch)",
                                   "achieved": achieved, "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "mean_stage_ms": msm_ms,
                                   "timing": "HIP events on the launch's stream, one launch in flight (after the timed region)" if isolated else "HIP events on the launch's stream, inside the timed region (other launches in flight)"},
                         "alu": {"note": "the stage is bound by 32-bit integer multiply issue, not by HBM: achieved Fq products/s of the stage against the "
                                         "measured chip-wide peak of the Montgomery product (tools/limb29_microbench.hip)",
                                 "fq_products_per_term": 2 * 12 * 11, "fq_products_per_term_note": "2 GLV halves x 12 windows (c = 11, the 1024-proof step) x 11 per mixed addition",
                                 "achieved_Gprod_s": (2 * 12 * 11 * terms_total) / (msm_ms * 1e-3) / 1e9 if msm_ms > 0 else 0.0,
                                 "peak_Gprod_s": 168.0}},
            "stages_ms": isolated if isolated else stages,
            "stages_ms_note": "HIP events between the stages, launches re-timed one at a time after the timed region (inside it only msm_accumulate is timed: an event between stages is a barrier packet, ~6 us of idle stream)",
            "stages_ms_one_launch_in_flight": isolated,
        }
        if tuning:
            out["tuning"] = tuning   # NOT the library's own choice of kernel variants: a measurement aid (--tuning)
        if m_re:
            out["value_reupload"] = m_re["total"] * args.steps / m_re["dt"]
            out["value_reupload_note"] = ("same K steps, host buffers (proofs, instances, draws) copied to the device again for every launch — PCIe-inclusive, SURVEY.md §8(d)'s "
                                          "timing method; the copy travels in chunks under the launch's point decompression (h2v_batch_upload_launch).  `value` is the HBM-resident figure "
                                          "the bench contract asks for; this one is what a caller that hands over host buffers per batch sees")
            out["value_reupload_ms_per_step_runs"] = [r / args.steps * 1e3 for r in m_re["runs"]]
        if m_c3:
            out["config3"] = {"workload": f"BASELINE.json configs 3/5: {8192 * world} proofs per step over {world} GPUs (8192 per GPU), one pairing per step after the RCCL all-gather; "
                                          f"{m_c3['G']} steps per launch, {m_c3['depth']} launches in flight, {args.config3_steps} steps timed",
                              "value": m_c3["total"] * args.config3_steps / m_c3["dt"], "unit": "proofs/s", "ms_per_step": m_c3["dt"] / args.config3_steps * 1e3,
                              "proofs_per_step": m_c3["total"]}
        if world == 1 and not args.no_extra_legs:
            out["config4"] = config4_leg(h2v, args, log)
            out["single_strategy"] = single_strategy_leg(h2v, ctx, d, args)
        if world == 1 and m["group0"] is not None and not args.no_cpu_baseline:
            # parity of the benched launch itself: timed group 0 (the accumulators of its last timed launch) against the CPU oracle
            # on the same proofs and the same draws — (left, right) bit for bit, both accepting
            reps = (B + args.distinct - 1) // args.distinct
            dd = {**d, "proofs": (d["proofs"] * reps)[: B * 1024], "inst": (d["inst"] * reps)[: B * 32 * N_PUBLIC]}
            t0 = time.perf_counter()
            ok_o, left_o, right_o = oracle_group(dd, B, m["group0_draws"][: 32 * B])
            match = ok_o and (left_o, right_o) == m["group0"]
            out["parity"] = {"checked": f"group 0 of the last timed launch ({B} proofs, its own draws): the (left, right) accumulator bytes of the GPU == the CPU oracle's, and both accept",
                             "ok": bool(match), "oracle_s": time.perf_counter() - t0}
            if not match:
                print(json.dumps(out), file=sys.stderr, flush=True)
                raise SystemExit("bench.py: the benched launch's accumulators differ from the CPU oracle's")
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline({**d, "proofs": (d["proofs"] * ((args.cpu_sample + args.distinct - 1) // args.distinct))[: args.cpu_sample * 1024],
                                                "inst": (d["inst"] * ((args.cpu_sample + args.distinct - 1) // args.distinct))[: args.cpu_sample * 32 * N_PUBLIC]},
                                               args.cpu_sample, log)
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2048)
    ap.add_argument("--warmup", type=int, default=256)
    ap.add_argument("--batch", type=int, default=1024, help="proofs per GPU per step")
    ap.add_argument("--groups", type=int, default=0, help="steps (independent batches, one pairing each) carried by one launch (h2v_batch_set_groups); 0 = choose from --steps")
    ap.add_argument("--depth", type=int, default=0, help="launches in flight per GPU (one HIP stream each); 0 = 8")
    ap.add_argument("--distinct", type=int, default=1024, help="distinct proofs generated (cycled if --batch is larger)")
    ap.add_argument("--cpu-sample", type=int, default=2048)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--reupload", action="store_true", help="copy the host buffers to the device again before every launch inside the headline run itself")
    ap.add_argument("--no-reupload-leg", action="store_true", help="skip the extra PCIe-inclusive leg (value_reupload)")
    ap.add_argument("--no-config3", action="store_true", help="N > 1: skip the extra 8192-proofs-per-GPU leg (BASELINE.json configs 3/5)")
    ap.add_argument("--config3-steps", type=int, default=8)
    ap.add_argument("--repeats", type=int, default=21, help="a timed region shorter than 50 ms is run this many times and the median reported (1 = never repeat; SURVEY.md §8(d): at least 10 WARM runs, median — the first six or seven runs of a process are a clock ramp, profiles/r03_repeat_curve.txt)")
    ap.add_argument("--tuning", default="", help="measurement aid: forced kernel variants, key=value[,key=value...] of h2v_tuning (default: automatic)")
    ap.add_argument("--no-extra-legs", action="store_true", help="N = 1: skip the config-4 (lookup-heavy VK) and SingleStrategy legs")
    ap.add_argument("--dry-run", action="store_true", help="print the launch plan (and, for N > 1, prove the N-rank rendezvous over gloo) without touching the GPU")
    args = ap.parse_args()
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0:
        raise SystemExit("bench.py: --gpus and --steps must be >= 1, --warmup >= 0")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process only launches the N ranks (it imports neither torch nor the HIP library)
        from halo2_verifier_amd.launch import spawn_ranks
        sys.exit(spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    rank_main(args)


if __name__ == "__main__":
    main()
