"""The N > 1 path on one GPU: R shards of one batch through halo2_verifier_amd.distributed (ShardedBatch /
verify_batch_sharded_local): each shard run through the staged C ABI with the draw tail of its global position
(h2v_batch_upload rand32_tail), accumulators exported and folded by h2v_batch_fold_check_enqueue — the call sequence
bench.py and verify_batch_sharded issue per rank, minus the all-gather (covered with two real ranks below over gloo, and on
CPU by tests/test_distributed_gloo.py).  The folded result must equal the unsharded h2v_verify_batch bit for bit, for any R.

Also the BASELINE.json-size checks (1024 proofs): determinism, sharding invariance, and single-proof tampering."""
import random

import pytest

import circuits
from circuits import R_MOD

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big():
    """1024 distinct proofs of the bench VK shape (k = 8 here: verifier work does not depend on k)."""
    s = circuits.setup_vector_mul(8, 8)
    P, I = circuits.prove_vector_mul_batch(s, 1024, seed=1234, threads=16)
    yield s, P, I
    s.free()


def _sharded(ctx, P, I, rand, R):
    """R shards on this one GPU through the library's entry point (distributed.verify_batch_sharded_local): the fold is done by
    shard 0; its verdict alone counts — the exchanged records carry every shard's failed-proof count (H2V_ACC_RECORD_BYTES), so no
    status AND across shards is needed on the host."""
    from halo2_verifier_amd import distributed as h2d
    return h2d.verify_batch_sharded_local(ctx, P, I, rand, R)


def _ctx(s):
    import halo2_verifier_amd as h2v
    return h2v.Context(h2v.ParamsKZG(s.params, h2v.SerdeFormat.RawBytes), h2v.VerifyingKey(s.vk, h2v.SerdeFormat.RawBytes))


@pytest.mark.parametrize("R", [1, 2, 3, 8])
def test_sharded_equals_unsharded(big, R):
    s, P, I = big
    ctx = _ctx(s)
    n = 100
    rnd = random.Random(R)
    rand = [rnd.randrange(1, R_MOD) for _ in range(n)]
    ref = ctx.verify_batch(P[:n], I[:n], rand)
    got = _sharded(ctx, P[:n], I[:n], rand, R)
    assert got == ref and got[0] is True
    ctx.close()


def test_failed_proof_on_a_non_folding_shard_rejects_the_batch(big):
    """A proof that fails BEFORE the MSM (undecodable point) is zeroed out of its shard's accumulators, so the folded pairing
    alone would pass.  The record a shard exports carries its failed-proof count; the rank that folds (shard 0 here, which
    holds only good proofs) must clear the verdict.  ADVICE r1, batch.hip finish_impl."""
    s, P, I = big
    ctx = _ctx(s)
    n = 24
    rnd = random.Random(99)
    rand = [rnd.randrange(1, R_MOD) for _ in range(n)]
    bad = list(P[:n])
    b = bytearray(bad[19]); b[0:32] = b"\xff" * 32; bad[19] = bytes(b)      # first advice commitment: x >= p, not decodable
    ref = ctx.verify_batch(bad, I[:n], rand)
    assert ref[0] is False and ref[1][19] == -5 and sum(1 for v in ref[1] if v) == 1
    assert circuits.oracle_verify_batch(s, bad, I[:n], rand) == ref
    for R in (2, 3, 4):
        got = _sharded(ctx, bad, I[:n], rand, R)      # proof 19 sits on the last shard; shard 0 folds
        assert got == ref, R
    # and the fold still accepts the untampered batch
    assert _sharded(ctx, P[:n], I[:n], rand, 3)[0] is True
    ctx.close()


def test_full_size_batch_properties(big):
    """1024 proofs (BASELINE.json config 2): determinism, sharding invariance at 8 shards, oracle agreement on the
    accumulators, and rejection + localisation of a single tampered proof."""
    s, P, I = big
    ctx = _ctx(s)
    rnd = random.Random(77)
    rand = [rnd.randrange(1, R_MOD) for _ in range(1024)]
    a = ctx.verify_batch(P, I, rand)
    b = ctx.verify_batch(P, I, rand)
    assert a == b and a[0] is True and a[1] == [0] * 1024
    assert _sharded(ctx, P, I, rand, 8) == a
    assert circuits.oracle_verify_batch(s, P, I, rand) == a            # bit-exact against the CPU oracle at full size (~2 s of CPU)
    # different draws, same proofs: still accepted, different accumulators
    rand2 = [rnd.randrange(1, R_MOD) for _ in range(1024)]
    c = ctx.verify_batch(P, I, rand2)
    assert c[0] is True and (c[2], c[3]) != (a[2], a[3])
    # one wrong public input among 1024: the batch is rejected; SingleStrategy pins it to the proof
    I2 = list(I)
    I2[517] = [[circuits.le32(1)] + I[517][0][1:]]
    d = ctx.verify_batch(P, I2, rand)
    assert d[0] is False and d[1] == [0] * 1024
    each = ctx.verify_each(P[512:520], I2[512:520])
    assert each == [0, 0, 0, 0, 0, -2, 0, 0]
    # OS-drawn multipliers (rand32 = NULL): accepted, and two runs disagree on the accumulators
    e1 = ctx.verify_batch(P[:64], I[:64], None)
    e2 = ctx.verify_batch(P[:64], I[:64], None)
    assert e1[0] and e2[0] and (e1[2], e1[3]) != (e2[2], e2[3])
    ctx.close()


def test_config3_shard_size(big):
    """BASELINE.json config 3 gives every GPU 8192 proofs (65 536 over 8).  At that size (the 1024 distinct proofs cycled):
    one batch == the fold of 8 shards of 1024 with the draw tails of their global positions; the same bytes as one grouped
    launch == 8 independent 1024-proof batches; and a single bad proof anywhere rejects the whole batch."""
    import torch
    import halo2_verifier_amd as h2v
    from halo2_verifier_amd import distributed as h2d
    s, P, I = big
    ctx = _ctx(s)
    n = 8192
    PP, II = (P * 8)[:n], (I * 8)[:n]
    rnd = random.Random(4242)
    rand = [rnd.randrange(1, R_MOD) for _ in range(n)]
    rand_all = b"".join(r.to_bytes(32, "little") for r in rand)
    flat = b"".join(PP)
    inst = b"".join(b"".join(col) for i in II for col in i)
    b = h2v.Batch(ctx, n, 8)
    b.upload(flat, 1024, inst, [8], rand_all)
    b.launch()
    whole = b.finish()
    b.close()
    assert whole[0] is True and whole[1] == [0] * n
    assert _sharded(ctx, PP, II, rand, 8) == whole
    # the same bytes as ONE grouped launch: 8 independent batches
    g = h2v.Batch(ctx, n, 8, groups=8)
    # (as independent batches: a group's multipliers stop at the end of its own 1024 proofs)
    g.upload(flat, 1024, inst, [8], rand_all)
    g.launch()
    oks, st, lefts, rights = g.finish_groups()
    g.close()
    assert oks == [True] * 8 and st == [0] * n
    for r in (0, 3, 7):
        sl = slice(1024 * r, 1024 * (r + 1))
        ref = ctx.verify_batch(PP[sl], II[sl], rand[sl])
        assert (oks[r], lefts[r], rights[r]) == (ref[0], ref[2], ref[3])
    # one tampered public input at position 5000
    I2 = list(II)
    I2[5000] = [[circuits.le32(7)] + II[5000][0][1:]]
    inst2 = b"".join(b"".join(col) for i in I2 for col in i)
    b = h2v.Batch(ctx, n, 8)
    b.upload(flat, 1024, inst2, [8], rand_all)
    b.launch()
    bad = b.finish()
    b.close()
    assert bad[0] is False and bad[1] == [0] * n
    ctx.close()


def test_config2_exactly_1024_proofs_at_k14():
    """BASELINE.json config 2 as stated: a batch of 1024 distinct proofs at k = 14 on one GPU, bit-exact against the oracle —
    statuses, both accumulator points and the verdict (the oracle needs ~2 s for the 1024 proofs).  The proofs are the bench's own
    (bench.load_or_make_proofs: cached under .bench_cache/, generated with the test prover if the cache is missing)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import halo2_verifier_amd as h2v
    d = bench.load_or_make_proofs(1024, 14, lambda m: None)
    P = [d["proofs"][1024 * i:1024 * (i + 1)] for i in range(1024)]
    I = [[[d["inst"][32 * (8 * i + j):32 * (8 * i + j + 1)] for j in range(8)]] for i in range(1024)]

    class S:   # what circuits.oracle_verify_batch needs
        L = circuits.oracle_lib.load(); params = d["params"]; vk = d["vk"]; ninst_cols = 1; multiopen = 0; transcript = 0
    assert int.from_bytes(d["params"][:4], "little") == 14
    ctx = h2v.Context(h2v.ParamsKZG(d["params"], h2v.SerdeFormat.RawBytes), h2v.VerifyingKey(d["vk"], h2v.SerdeFormat.RawBytes))
    rnd = random.Random(1414)
    rand = [rnd.randrange(1, R_MOD) for _ in range(1024)]
    got = ctx.verify_batch(P, I, rand)
    exp = circuits.oracle_verify_batch(S, P, I, rand)
    assert got == exp and got[0] is True and got[1] == [0] * 1024
    # one flipped byte in proof 700's last evaluation: Transcript-clean, rejected by the pairing, same accumulators as the oracle
    b = bytearray(P[700]); b[900] ^= 1; P2 = list(P); P2[700] = bytes(b)
    got = ctx.verify_batch(P2, I, rand)
    assert got == circuits.oracle_verify_batch(S, P2, I, rand) and got[0] is False
    ctx.close()


def test_rccl_backend_single_rank_smoke(big):
    """The collective call the N > 1 bench issues — init_process_group("nccl") (= RCCL) with a device id, all_gather_into_tensor of
    uint8 accumulator records on a side stream, fold + pairing enqueued behind it — with a world of ONE rank: the only RCCL run
    this one-GPU pool allows (RCCL refuses two ranks on one device).  It pins the dtype / stream / API usage, not the exchange."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    import halo2_verifier_amd as h2v
    from halo2_verifier_amd import distributed as h2d
    s, P, I = big
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        ctx = _ctx(s)
        n, G = 32, 2
        rnd = random.Random(8)
        rand = [rnd.randrange(1, R_MOD) for _ in range(n)]
        stream = torch.cuda.Stream()
        b = h2v.Batch(ctx, n, 8, stream=stream.cuda_stream, groups=G)
        flat = b"".join(P[:n]); inst = b"".join(b"".join(col) for i in I[:n] for col in i)
        b.upload(flat, 1024, inst, [8], b"".join(r.to_bytes(32, "little") for r in rand))
        local = torch.empty(h2d.ACC_BYTES * G, dtype=torch.uint8, device="cuda:0")
        with torch.cuda.stream(stream):
            b.launch(with_pairing=False)
            b.export_accumulators(local.data_ptr())
            out = torch.empty_like(local)
            dist.all_gather_into_tensor(out, local)          # what gather_accumulators does for world_size > 1
            b.fold_check_enqueue(out.data_ptr(), 1)
        ok, st, left, right = b.finish_groups()
        dist.barrier(device_ids=[0])
        assert ok == [True, True] and st == [0] * n
        for g in range(G):
            ref = ctx.verify_batch(P[g * 16:(g + 1) * 16], I[g * 16:(g + 1) * 16], rand[g * 16:(g + 1) * 16])
            assert (left[g], right[g]) == (ref[2], ref[3]) and ref[0]
        b.close(); ctx.close()
    finally:
        dist.destroy_process_group()


def test_config3_and_5_full_size_through_the_entry_point(big):
    """BASELINE.json configs 3 and 5 AS STATED, on one GPU: 65 536 proofs (the 1024 distinct ones cycled) as 8 shards of 8192 with the
    draw tails of their global positions, the 8 records folded, ONE pairing — equal bit for bit to the unsharded 65 536-proof
    h2v_verify_batch; and a single proof rejected on a NON-folding shard (shard 5) rejects the whole batch."""
    from halo2_verifier_amd import distributed as h2d
    s, P, I = big
    ctx = _ctx(s)
    n = 65536
    Pn = [P[i % 1024] for i in range(n)]
    In = [I[i % 1024] for i in range(n)]
    rnd = random.Random(65536)
    rand = [rnd.randrange(1, R_MOD) for _ in range(n)]
    ref = ctx.verify_batch(Pn, In, rand)
    assert ref[0] is True and ref[1] == [0] * n
    got = h2d.verify_batch_sharded_local(ctx, Pn, In, rand, 8)
    assert got == ref
    # the same unsharded result from two very different launch shapes: one batch, and the fold of 64 shards of 1024
    assert h2d.verify_batch_sharded_local(ctx, Pn, In, rand, 64) == ref
    bad = list(Pn)
    k = 5 * 8192 + 4097
    b = bytearray(bad[k]); b[64:96] = b"\xff" * 32; bad[k] = bytes(b)
    got = h2d.verify_batch_sharded_local(ctx, bad, In, rand, 8)
    assert got[0] is False and [i for i, v in enumerate(got[1]) if v] == [k] and got[1][k] == -5
    assert got == ctx.verify_batch(bad, In, rand)
    ctx.close()


def _rank_main(rank, world, port, q):
    import os
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import circuits as C
    import halo2_verifier_amd as h2v
    from halo2_verifier_amd import distributed as h2d
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)     # RCCL refuses two ranks on one device: gloo carries the records
    s = C.setup_vector_mul(8, 8)
    P, I = C.prove_vector_mul_batch(s, 37, seed=99, threads=4)
    ctx = h2v.Context(h2v.ParamsKZG(s.params, h2v.SerdeFormat.RawBytes), h2v.VerifyingKey(s.vk, h2v.SerdeFormat.RawBytes), device=0)
    rnd = random.Random(3)
    rand = [rnd.randrange(1, C.R_MOD) for _ in range(37)]
    good = h2d.verify_batch_sharded(ctx, P, I, rand)
    bad = list(P); b = bytearray(bad[30]); b[0:32] = b"\xff" * 32; bad[30] = bytes(b)       # on the last rank's shard
    rej = h2d.verify_batch_sharded(ctx, bad, I, rand)
    drawn = h2d.verify_batch_sharded(ctx, P, I, None)                  # rank 0 draws and broadcasts
    tiny = h2d.verify_batch_sharded(ctx, P[:1], I[:1], rand[:1])       # the last rank's shard is empty
    ref = (ctx.verify_batch(P, I, rand), ctx.verify_batch(bad, I, rand), ctx.verify_batch(P[:1], I[:1], rand[:1])) if rank == 0 else None
    q.put((rank, good, rej, drawn, tiny, ref))
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


@pytest.mark.parametrize("world", [2, 3])
def test_verify_batch_sharded_with_real_ranks_on_one_gpu(world):
    """distributed.verify_batch_sharded with `world` real ranks (processes), all on cuda:0, records over gloo: every rank returns the
    unsharded h2v_verify_batch result — accepted batch, a proof rejected on another rank's shard, OS-drawn multipliers broadcast by
    rank 0 (every rank must report the same accumulators), and a batch smaller than the world (empty shards)."""
    import socket
    import torch.multiprocessing as mp
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_rank_main, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda t: t[0])
    for p in procs: p.join(120)
    assert all(p.exitcode == 0 for p in procs)
    ref = res[0][5]
    for rank, good, rej, drawn, tiny, _ in res:
        assert good == ref[0] and good[0] is True
        assert rej == ref[1] and rej[0] is False and rej[1][30] == -5
        assert tiny == ref[2] and tiny[0] is True
        assert drawn[0] is True and drawn == res[0][3]
