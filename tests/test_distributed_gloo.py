"""The N > 1 path on CPU: world_size-2 gloo run of the sharding logic in halo2_verifier_amd.distributed.

What is exercised: shard_bounds / tail_for_shard / gather_accumulators (the same all_gather_into_tensor call the
GPU path issues over RCCL), and the algebra that makes the sharded result independent of the number of ranks:
proof i is scaled by the product of the draws of ALL later proofs of the whole batch (kzg/strategy.rs:129,
msm.rs:173-176), so a shard's accumulator computed from its own draws only needs the product of the later
shards' draws — which is exactly what uploading the draw tail [lo, total) to h2v_batch_upload gives the GPU.
The per-shard group arithmetic is done by the CPU oracle here (there is no GPU in this container); the GPU
counterpart is tests/test_gpu_sharded.py."""
import os
import random
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R_MOD = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import circuits
    import oracle_lib
    from halo2_verifier_amd import distributed as h2d
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = oracle_lib.load()
    s = circuits.setup_vector_mul(8, 10)
    P, I = circuits.prove_vector_mul_batch(s, total, seed=5, threads=2)
    rnd = random.Random(2024)
    rand = [rnd.randrange(1, R_MOD) for _ in range(total)]
    rand_all = b"".join(r.to_bytes(32, "little") for r in rand)
    lo, hi = h2d.shard_bounds(total, world, rank)
    tail = h2d.tail_for_shard(rand_all, lo)
    assert len(tail) == 32 * (total - lo)
    # shard accumulator with its own draws, then scaled by the product of the later shards' draws (the tail beyond hi)
    ok, st, left, right = circuits.oracle_verify_batch(s, P[lo:hi], I[lo:hi], rand[lo:hi])
    T = 1
    for j in range(hi, total):
        T = T * rand[j] % R_MOD
    left = oracle_lib.g1_msm(L, [T], [left]); right = oracle_lib.g1_msm(L, [T], [right])
    local = torch.zeros(h2d.ACC_BYTES, dtype=torch.uint8)
    local[:128] = torch.tensor(list(left + right), dtype=torch.uint8)
    gathered = h2d.gather_accumulators(local, world)
    parts = [bytes(gathered[i * h2d.ACC_BYTES:i * h2d.ACC_BYTES + 128].tolist()) for i in range(world)]
    fl = oracle_lib.g1_msm(L, [1] * world, [p[:64] for p in parts])
    fr = oracle_lib.g1_msm(L, [1] * world, [p[64:] for p in parts])
    if rank == 0:
        full = circuits.oracle_verify_batch(s, P, I, rand)
        q.put((fl == full[2], fr == full[3], full[0]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [5, 8])
def test_two_rank_sharding_matches_unsharded(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs: p.start()
    for p in procs: p.join(300)
    assert all(p.exitcode == 0 for p in procs)
    assert q.get(timeout=5) == (True, True, True)


def test_shard_bounds_cover_everything():
    from halo2_verifier_amd.distributed import shard_bounds
    for total in (0, 1, 7, 8, 1024, 65536):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def _worker_groups(rank, world, port, G, q):
    sys.path.insert(0, ROOT)
    from halo2_verifier_amd import distributed as h2d
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # what Batch.export_accumulators writes for a grouped launch: [group][left, right] opaque bytes
    local = torch.tensor([(rank * 131 + g * 17 + k) % 251 for g in range(G) for k in range(h2d.ACC_BYTES)], dtype=torch.uint8)
    gathered = h2d.gather_accumulators(local, world)
    ok = gathered.numel() == world * G * h2d.ACC_BYTES
    for r in range(world):
        for g in range(G):
            off = (r * G + g) * h2d.ACC_BYTES     # the [rank][group] layout h2v_batch_fold_check_enqueue folds
            want = [(r * 131 + g * 17 + k) % 251 for k in range(h2d.ACC_BYTES)]
            ok = ok and gathered[off:off + h2d.ACC_BYTES].tolist() == want
    if rank == 0:
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_grouped_accumulators_gather_in_rank_major_order():
    """A grouped launch exchanges all its groups in ONE collective; the fold kernel indexes the result as [rank][group]."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_groups, args=(r, 2, port, 5, q)) for r in range(2)]
    for p in procs: p.start()
    for p in procs: p.join(120)
    assert all(p.exitcode == 0 for p in procs)
    assert q.get(timeout=5) is True


def _worker_entry(rank, world, port, total, tamper, draw_locally, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import circuits
    import fake_shard
    from halo2_verifier_amd import distributed as h2d
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = circuits.setup_vector_mul(8, 10)
    P, I = circuits.prove_vector_mul_batch(s, total, seed=6, threads=2)
    if tamper >= 0:
        b = bytearray(P[tamper]); b[0:32] = b"\xff" * 32; P[tamper] = bytes(b)      # undecodable point: the proof leaves its shard's accumulators
    rnd = random.Random(77)
    rand = None if draw_locally else [rnd.randrange(1, R_MOD) for _ in range(total)]

    class Ctx:                      # the stand-in never touches it
        device = 0
    got = h2d.verify_batch_sharded(Ctx(), P, I, rand, batch_factory=fake_shard.make_factory(s), device="cpu")
    if rand is None:
        # rank 0 drew and broadcast: all ranks must have used the same stream, i.e. report the same accumulators
        blob = torch.tensor(list(got[2] + got[3]), dtype=torch.uint8)
        both = [torch.zeros_like(blob) for _ in range(world)]
        dist.all_gather(both, blob)
        same = all(bool((b == both[0]).all()) for b in both)
        if rank == 0:
            q.put((got[0], got[1], same))
    else:
        ref = circuits.oracle_verify_batch(s, P, I, rand)
        one = h2d.verify_batch_sharded_local(Ctx(), P, I, rand, 3, batch_factory=fake_shard.make_factory(s), device="cpu")
        q.put((rank, got == ref, one == ref, got[0]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total,tamper", [(7, -1), (6, 4), (1, -1)])
def test_verify_batch_sharded_entry_point_two_ranks(total, tamper):
    """distributed.verify_batch_sharded — the product's entry point for a sharded batch — over gloo with two ranks (the device side
    replaced by tests/fake_shard.py: there is no GPU here): every rank returns the unsharded oracle result, a proof rejected on
    one shard rejects the batch on every rank, a shard may be empty (1 proof, 2 ranks), and the one-device
    sequential form (verify_batch_sharded_local, 3 shards) agrees."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_entry, args=(r, 2, port, total, tamper, False, q)) for r in range(2)]
    for p in procs: p.start()
    for p in procs: p.join(300)
    assert all(p.exitcode == 0 for p in procs)
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert [r[0] for r in res] == [0, 1]
    assert all(r[1] and r[2] for r in res)
    assert all(r[3] is (tamper < 0) for r in res)


def test_verify_batch_sharded_draws_are_common_when_rank0_draws():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_entry, args=(r, 2, port, 5, -1, True, q)) for r in range(2)]
    for p in procs: p.start()
    for p in procs: p.join(300)
    assert all(p.exitcode == 0 for p in procs)
    ok, st, same = q.get(timeout=5)
    assert ok is True and st == [0] * 5 and same is True
