"""Grouped batches (h2v_batch_set_groups / h2v_batch_finish_groups): G independent AccumulatorStrategy batches in one
upload / launch.  Every group must give bit for bit what a separate h2v_verify_batch over its proofs and its slice of
the draws gives — and what the CPU oracle gives — including when one group holds a bad proof, when the groups are
sharded over ranks (export + fold per group), for the GWC / Keccak plan, and at the bench's group shape."""
import random

import pytest

import circuits
from circuits import R_MOD

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pool():
    s = circuits.setup_vector_mul(8, 8)
    P, I = circuits.prove_vector_mul_batch(s, 256, seed=4321, threads=16)
    yield s, P, I
    s.free()


def _ctx(s):
    import halo2_verifier_amd as h2v
    return h2v.Context(h2v.ParamsKZG(s.params, h2v.SerdeFormat.RawBytes), h2v.VerifyingKey(s.vk, h2v.SerdeFormat.RawBytes))


def _flat(P, I):
    return b"".join(P), b"".join(b"".join(col) for i in I for col in i)


def _rand_bytes(rand):
    return b"".join(r.to_bytes(32, "little") for r in rand)


def _grouped(ctx, P, I, rand, G, with_pairing=True):
    import halo2_verifier_amd as h2v
    b = h2v.Batch(ctx, len(P), 8, groups=G)
    flat, inst = _flat(P, I)
    b.upload(flat, 1024, inst, [8], _rand_bytes(rand))
    b.launch(with_pairing=with_pairing)
    out = b.finish_groups()
    b.close()
    return out


@pytest.mark.parametrize("G,gs", [(1, 24), (2, 16), (5, 7), (8, 32), (16, 3)])
def test_groups_equal_separate_batches(pool, G, gs):
    s, P, I = pool
    ctx = _ctx(s)
    n = G * gs
    rnd = random.Random(100 * G + gs)
    rand = [rnd.randrange(1, R_MOD) for _ in range(n)]
    ok, st, left, right = _grouped(ctx, P[:n], I[:n], rand, G)
    assert st == [0] * n and ok == [True] * G
    for g in range(G):
        sl = slice(g * gs, (g + 1) * gs)
        ref = ctx.verify_batch(P[sl], I[sl], rand[sl])
        assert (ok[g], left[g], right[g]) == (ref[0], ref[2], ref[3])
        if g in (0, G - 1):
            assert circuits.oracle_verify_batch(s, P[sl], I[sl], rand[sl]) == ref
    assert len(set(left)) == G  # the groups really are distinct batches
    ctx.close()


def test_bad_proof_fails_only_its_group(pool):
    s, P, I = pool
    ctx = _ctx(s)
    G, gs = 4, 8
    n = G * gs
    rnd = random.Random(9)
    rand = [rnd.randrange(1, R_MOD) for _ in range(n)]
    # group 1: wrong public input (pairing fails, no per-proof status); group 3: an undecodable opening point (per-proof status)
    I2 = list(I[:n])
    I2[gs + 3] = [[circuits.le32(5)] + I[gs + 3][0][1:]]
    P2 = list(P[:n])
    bad = bytearray(P2[3 * gs + 1]); bad[-33] = 0xff; P2[3 * gs + 1] = bytes(bad)   # top byte of h1: x >= p
    ok, st, left, right = _grouped(ctx, P2, I2, rand, G)
    assert ok == [True, False, True, False]
    assert [i for i, v in enumerate(st) if v] == [3 * gs + 1]
    for g in range(G):
        sl = slice(g * gs, (g + 1) * gs)
        ref = ctx.verify_batch(P2[sl], I2[sl], rand[sl])
        assert (ok[g], st[sl], left[g], right[g]) == ref
        assert circuits.oracle_verify_batch(s, P2[sl], I2[sl], rand[sl]) == ref
    ctx.close()


def test_grouped_and_sharded(pool):
    """G global batches, each split over R ranks: rank r holds proofs [r*gs, (r+1)*gs) of every group and uploads, per
    group, the draws from its first proof to the end of the group's global batch; the exported accumulators of all ranks
    ([rank][group][left, right]) are folded group by group.  Equal to G unsharded h2v_verify_batch calls."""
    import torch
    import halo2_verifier_amd as h2v
    s, P, I = pool
    ctx = _ctx(s)
    G, R, gs = 3, 4, 5            # global batch of a group: R * gs = 20 proofs
    per_group = R * gs
    rnd = random.Random(31)
    rand = [[rnd.randrange(1, R_MOD) for _ in range(per_group)] for _ in range(G)]
    proofs = [P[g * per_group:(g + 1) * per_group] for g in range(G)]
    insts = [I[g * per_group:(g + 1) * per_group] for g in range(G)]
    from halo2_verifier_amd import distributed as h2d
    acc_bytes = h2d.ACC_BYTES * G
    acc = torch.zeros(R * acc_bytes, dtype=torch.uint8, device="cuda:0")
    batches = []
    for r in range(R):
        b = h2v.Batch(ctx, G * gs, 8, groups=G)
        Pl = [p for g in range(G) for p in proofs[g][r * gs:(r + 1) * gs]]
        Il = [i for g in range(G) for i in insts[g][r * gs:(r + 1) * gs]]
        tail = b"".join(_rand_bytes(rand[g][r * gs:]) for g in range(G))
        flat, inst = _flat(Pl, Il)
        b.upload(flat, 1024, inst, [8], tail)
        b.launch(with_pairing=False)
        b.export_accumulators(acc.data_ptr() + r * acc_bytes)
        okg, st, _, _ = b.finish_groups()
        assert st == [0] * (G * gs)
        batches.append(b)
    torch.cuda.synchronize()
    batches[0].fold_check_enqueue(acc.data_ptr(), R)
    ok, _, left, right = batches[0].finish_groups()
    for g in range(G):
        ref = ctx.verify_batch(proofs[g], insts[g], rand[g])
        assert (ok[g], left[g], right[g]) == (ref[0], ref[2], ref[3]) and ref[0]
    for b in batches:
        b.close()
    ctx.close()


def test_group_argument_checks(pool):
    import halo2_verifier_amd as h2v
    s, P, I = pool
    ctx = _ctx(s)
    b = h2v.Batch(ctx, 12, 8, groups=4)
    flat, inst = _flat(P[:10], I[:10])
    with pytest.raises(h2v.H2VError):       # 10 proofs do not split into 4 equal groups
        b.upload(flat, 1024, inst, [8], _rand_bytes([1] * 10))
    flat, inst = _flat(P[:12], I[:12])
    b.upload(flat, 1024, inst, [8], _rand_bytes([1] * 12))
    b.launch()
    with pytest.raises(h2v.H2VError):       # a grouped batch has no single verdict
        b.finish()
    ok, st, _, _ = b.finish_groups()
    assert ok == [True] * 4 and st == [0] * 12
    with pytest.raises(h2v.H2VError):
        b.set_groups(0)
    # back to one group: the plain finish works again on the same object
    b.set_groups(1)
    b.upload(flat, 1024, inst, [8], _rand_bytes([3] * 12))
    b.launch()
    ok1, st1, l1, r1 = b.finish()
    assert (ok1, st1, l1, r1) == ctx.verify_batch(P[:12], I[:12], [3] * 12)
    b.close()
    ctx.close()


def test_groups_gwc_keccak():
    s = circuits.setup_vector_mul(8, 4).set_options(circuits.GWC, circuits.KECCAK256)
    P, I = circuits.prove_vector_mul_batch(s, 12, seed=5, threads=4)
    import halo2_verifier_amd as h2v
    ctx = h2v.Context(h2v.ParamsKZG(s.params, h2v.SerdeFormat.RawBytes), h2v.VerifyingKey(s.vk, h2v.SerdeFormat.RawBytes),
                      multiopen=s.multiopen, transcript=s.transcript)
    rnd = random.Random(3)
    rand = [rnd.randrange(1, R_MOD) for _ in range(12)]
    b = h2v.Batch(ctx, 12, 4, groups=3)
    plen = len(P[0])
    flat, inst = _flat(P, I)
    b.upload(flat, plen, inst, [4], _rand_bytes(rand))
    b.launch()
    ok, st, left, right = b.finish_groups()
    b.close()
    assert ok == [True] * 3
    for g in range(3):
        sl = slice(4 * g, 4 * g + 4)
        ref = ctx.verify_batch(P[sl], I[sl], rand[sl])
        assert (ok[g], left[g], right[g]) == (ref[0], ref[2], ref[3])
        assert circuits.oracle_verify_batch(s, P[sl], I[sl], rand[sl]) == ref
    ctx.close()
    s.free()


def test_results_of_a_launch_whose_tail_runs_on_the_auxiliary_stream(pool):
    """A launch with its own pairing ends on two streams (the pairing kernel writes the verdicts into pinned host memory; the accumulator
    bytes and statuses come back on the auxiliary stream, not joined into the batch's stream): finish must wait for both, a second finish
    must return the same, and calls made BEFORE finish — another launch, an upload + launch of different proofs, an export — must order
    themselves behind that tail."""
    import torch
    import halo2_verifier_amd as h2v
    from halo2_verifier_amd.distributed import ACC_BYTES
    s, P, I = pool
    ctx = _ctx(s)
    G, gs = 4, 16
    n = G * gs
    rnd = random.Random(77)
    rand = [rnd.randrange(1, R_MOD) for _ in range(n)]
    rand2 = [rnd.randrange(1, R_MOD) for _ in range(n)]
    want = _grouped(ctx, P[:n], I[:n], rand, G)
    # the second set of proofs: group 1 holds a proof whose instances do not match
    P2, I2 = P[n:2 * n], [list(map(list, i)) for i in I[n:2 * n]]
    I2[gs + 3][0][0] = (5).to_bytes(32, "little")
    want2 = _grouped(ctx, P2, I2, rand2, G)
    assert want[0] == [True] * G and want2[0] == [True, False, True, True]

    b = h2v.Batch(ctx, n, 8, groups=G)
    flat, inst = _flat(P[:n], I[:n])
    flat2, inst2 = _flat(P2, I2)
    b.upload(flat, 1024, inst, [8], _rand_bytes(rand))
    b.launch(True)
    assert b.finish_groups() == want
    assert b.finish_groups() == want                     # nothing new enqueued: the same block again
    b.launch(True); b.launch(True)                       # a launch on top of an unfinished one
    assert b.finish_groups() == want
    b.launch(True)
    rec = torch.zeros(G * ACC_BYTES, dtype=torch.uint8, device="cuda:0")
    b.export_accumulators(rec.data_ptr())                # reads the whole accumulators the auxiliary stream puts together
    got = b.finish_groups()
    assert got == want
    whole = h2v.Batch(ctx, n, 8, groups=G)
    whole.upload(flat, 1024, inst, [8], _rand_bytes(rand))
    whole.launch(False)
    rec2 = torch.zeros_like(rec)
    whole.export_accumulators(rec2.data_ptr())
    whole.finish_groups()
    # (a launch without a pairing exports its pieces, one with a pairing has put the whole points together: the folded check must agree)
    for r_ in (rec, rec2):
        chk = h2v.Batch(ctx, n, 8, groups=G)
        chk.upload(flat, 1024, inst, [8], _rand_bytes(rand))
        chk.launch(False)
        chk.fold_check_enqueue(r_.data_ptr(), 1)
        okf, _, leftf, rightf = chk.finish_groups()
        assert (okf, leftf, rightf) == (want[0], want[2], want[3])
        chk.close()
    b.launch(True)
    b.upload(flat2, 1024, inst2, [8], _rand_bytes(rand2))   # new inputs while the previous launch's tail may still run
    b.launch(True)
    assert b.finish_groups() == want2
    b.upload_launch(flat, 1024, inst, [8], _rand_bytes(rand))
    assert b.finish_groups() == want
    b.close(); whole.close()
