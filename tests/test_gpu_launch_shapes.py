"""The kernel variants the library picks by LAUNCH SHAPE, at the shapes that select them — bit for bit.

libh2v_amd.so chooses kernels from the size of a launch (csrc/verify_kernels.hip frvm_enqueue, csrc/msm.hip msm_enqueue_multi,
csrc/batch.hip): the throughput launches of bench.py run variants that the small launches of the rest of the suite never reach
(round-2 review):
  * the Fr program as K = 2 instruction streams (> 341 waves = > 21 824 proofs per launch) with a 78 KB (<= 384 waves) or 36 KB
    LDS slice of its slot file — small launches run K = 4 with every slot in LDS;
  * msm_window with one wave per window (> 512 windows) and, beyond 1024 windows, its 20 KB form without the two-bit digit table;
  * more than 64 groups per launch: whole accumulators (full Horner) and the whole-point pairing instead of pieces;
  * h2v_verify_each beyond 512 proofs: several grouped launches of one-proof groups.
Every group of a grouped launch must equal — verdict, per-proof statuses, both accumulator points — a separate h2v_verify_batch
over the same proofs and draws, which tests/test_gpu_sharded.py ties to the CPU oracle at 1024 proofs (and two groups are checked
against the oracle here directly)."""
import random

import pytest

import circuits
from circuits import R_MOD

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pool():
    """1024 distinct proofs of the bench VK shape (k = 8: verifier work does not depend on k)."""
    s = circuits.setup_vector_mul(8, 8)
    P, I = circuits.prove_vector_mul_batch(s, 1024, seed=20261, threads=16)
    yield s, P, I
    s.free()


def _ctx(s):
    import halo2_verifier_amd as h2v
    return h2v.Context(h2v.ParamsKZG(s.params, h2v.SerdeFormat.RawBytes), h2v.VerifyingKey(s.vk, h2v.SerdeFormat.RawBytes))


def _flat(P, I):
    return b"".join(P), b"".join(b"".join(col) for i in I for col in i)


def _group_inputs(P, I, G, gs, seed):
    """G groups of gs proofs cycled out of the pool (group g starts at proof 7 g), their own draws, and two spoiled groups:
    one with a wrong public input (its pairing fails, no per-proof status) and one with an undecodable point (a per-proof status,
    the proof leaves the accumulators)."""
    rnd = random.Random(seed)
    n = len(P)
    groups = []
    for g in range(G):
        idx = [(7 * g + j) % n for j in range(gs)]
        groups.append(([P[i] for i in idx], [I[i] for i in idx], [rnd.randrange(1, R_MOD) for _ in range(gs)]))
    g_bad_inst, g_bad_point = G // 3, (2 * G) // 3
    Pg, Ig, r = groups[g_bad_inst]
    Ig = list(Ig); Ig[gs // 2] = [[circuits.le32(12345)] + Ig[gs // 2][0][1:]]
    groups[g_bad_inst] = (Pg, Ig, r)
    Pg, Ig, r = groups[g_bad_point]
    Pg = list(Pg); b = bytearray(Pg[gs - 3]); b[0:32] = b"\xff" * 32; Pg[gs - 3] = bytes(b)
    groups[g_bad_point] = (Pg, Ig, r)
    return groups, g_bad_inst, g_bad_point


# (groups, proofs per group): waves of the Fr program = G gs / 64; windows of the launch = 2 G x 12
@pytest.mark.parametrize("G,gs", [(24, 1024),    # 384 waves: K = 2, 78 KB slice; 576 windows: one wave per window, digit table
                                  (44, 1024),    # 704 waves: K = 2, 36 KB slice; 1056 windows: one wave per window, NO digit table (slots = 3)
                                  (66, 256)])    # more than 64 groups: whole accumulators, whole-point pairing (264 waves: K = 4)
def test_throughput_launch_groups_equal_separate_batches(pool, G, gs):
    import halo2_verifier_amd as h2v
    s, P, I = pool
    ctx = _ctx(s)
    groups, g_bad_inst, g_bad_point = _group_inputs(P, I, G, gs, 1000 * G + gs)
    b = h2v.Batch(ctx, G * gs, 8, groups=G)
    flat, inst = _flat([p for Pg, _, _ in groups for p in Pg], [i for _, Ig, _ in groups for i in Ig])
    b.upload(flat, 1024, inst, [8], b"".join(r.to_bytes(32, "little") for _, _, rg in groups for r in rg))
    b.launch(with_pairing=True)
    ok, st, left, right = b.finish_groups()
    b.close()
    assert [g for g in range(G) if not ok[g]] == sorted([g_bad_inst, g_bad_point])
    assert [i for i, v in enumerate(st) if v] == [g_bad_point * gs + gs - 3] and st[g_bad_point * gs + gs - 3] == -5
    for g, (Pg, Ig, rg) in enumerate(groups):
        ref = ctx.verify_batch(Pg, Ig, rg)      # a 1024-proof launch of its own: K = 4, all slots in LDS, four waves per window, pieces
        assert (ok[g], st[g * gs:(g + 1) * gs], left[g], right[g]) == ref, g
        if g in (0, g_bad_point):
            assert circuits.oracle_verify_batch(s, Pg, Ig, rg) == ref
    assert len(set(left)) == G
    ctx.close()


def test_throughput_launch_sharded_records(pool):
    """The same throughput-shaped launch WITHOUT its own pairing (what a rank of a sharded job runs): exported records folded with
    a second shard's, one pairing per group — equal to the unsharded batches."""
    import torch
    import halo2_verifier_amd as h2v
    from halo2_verifier_amd import distributed as h2d
    s, P, I = pool
    ctx = _ctx(s)
    G, gs, R = 44, 512, 2          # per group a global batch of R x gs = 1024 proofs; rank r holds its proofs [r gs, (r + 1) gs)
    rnd = random.Random(4)
    draws = [[rnd.randrange(1, R_MOD) for _ in range(R * gs)] for _ in range(G)]
    glob = [[(11 * g + j) % len(P) for j in range(R * gs)] for g in range(G)]
    acc = torch.zeros(R * G * h2d.ACC_BYTES, dtype=torch.uint8, device="cuda:0")
    batches = []
    for r in range(R):
        b = h2v.Batch(ctx, G * gs, 8, groups=G)
        idx = [i for g in range(G) for i in glob[g][r * gs:(r + 1) * gs]]
        flat, inst = _flat([P[i] for i in idx], [I[i] for i in idx])
        b.upload(flat, 1024, inst, [8], b"".join(x.to_bytes(32, "little") for g in range(G) for x in draws[g][r * gs:]))
        b.launch(with_pairing=False)
        b.export_accumulators(acc.data_ptr() + r * G * h2d.ACC_BYTES)
        okg, st, _, _ = b.finish_groups()
        assert st == [0] * (G * gs)
        batches.append(b)
    torch.cuda.synchronize()
    batches[0].fold_check_enqueue(acc.data_ptr(), R)
    ok, _, left, right = batches[0].finish_groups()
    for b in batches:
        b.close()
    assert ok == [True] * G
    for g in (0, 1, 21, 43):
        ref = ctx.verify_batch([P[i] for i in glob[g]], [I[i] for i in glob[g]], draws[g])
        assert (ok[g], left[g], right[g]) == (ref[0], ref[2], ref[3]), g
    ctx.close()


def test_verify_each_beyond_one_launch(pool):
    """SingleStrategy over 1100 proofs: three grouped launches of one-proof groups (512 + 512 + 76), a few spoiled proofs on both
    sides of the launch boundaries; statuses equal the oracle's single-proof verdicts."""
    s, P, I = pool
    ctx = _ctx(s)
    n = 1100
    Pn = [P[i % len(P)] for i in range(n)]
    In = [I[i % len(I)] for i in range(n)]
    expect = [0] * n
    for i in (0, 511, 512, 777, 1023, 1024, 1099):          # wrong public input: the proof's own pairing fails (kzg/strategy.rs:171-175)
        In[i] = [[circuits.le32(3 + i)] + In[i][0][1:]]
        expect[i] = -2
    for i in (5, 513, 1090):                                # undecodable point in the main part of the transcript
        b = bytearray(Pn[i]); b[32:64] = b"\xff" * 32; Pn[i] = bytes(b)
        expect[i] = -5
    for i in (600,):                                        # undecodable opening point
        b = bytearray(Pn[i]); b[-33] = 0xff; Pn[i] = bytes(b)
        expect[i] = -4
    got = ctx.verify_each(Pn, In)
    assert got == expect
    for i in (0, 5, 512, 513, 600, 1023, 1024, 1099, 2, 700):
        assert circuits.oracle_verify_single(s, Pn[i], In[i]) == expect[i], i
    ctx.close()


@pytest.mark.parametrize("G,gs,proof_len", [(1, 37, 1024), (4, 1024, 1024), (20, 1024, 1024), (3, 500, 1056)])
def test_upload_launch_equals_upload_then_launch(pool, G, gs, proof_len):
    """h2v_batch_upload_launch (the host -> device copy chunked under the point decompression) gives what upload + launch give:
    one chunk (small batches), several chunks, the driver's 20 x 1024 shape, proofs with trailing bytes (strided copy); a spoiled
    proof in the last chunk keeps its status; and a plain re-launch afterwards runs every stage again."""
    import halo2_verifier_amd as h2v
    s, P, I = pool
    ctx = _ctx(s)
    rnd = random.Random(G * 7 + gs)
    n = G * gs
    idx = [(13 * i) % len(P) for i in range(n)]
    Pn = [P[i] + bytes(proof_len - 1024) for i in idx]
    In = [I[i] for i in idx]
    bad = bytearray(Pn[n - 2]); bad[32:64] = b"\xff" * 32; Pn[n - 2] = bytes(bad)
    rand = b"".join(rnd.randrange(1, R_MOD).to_bytes(32, "little") for _ in range(n))
    flat, inst = _flat(Pn, In)
    a = h2v.Batch(ctx, n, 8, groups=G)
    a.upload(flat, proof_len, inst, [8], rand)
    a.launch(with_pairing=True)
    ref = a.finish_groups()
    a.close()
    b = h2v.Batch(ctx, n, 8, groups=G)
    b.upload_launch(flat, proof_len, inst, [8], rand, with_pairing=True)
    got = b.finish_groups()
    assert got == ref and got[1][n - 2] == -5 and got[0][:-1] == [True] * (G - 1) and got[0][-1] is False
    b.launch(with_pairing=True)                 # the same upload launched again: decompression included
    assert b.finish_groups() == ref
    # a second upload_launch into the same object with other inputs (buffers reused, events reused)
    good = bytes(Pn[n - 2][:32]) + P[idx[n - 2]][32:] + bytes(proof_len - 1024)
    flat2 = flat[: (n - 2) * proof_len] + P[idx[n - 2]] + bytes(proof_len - 1024) + flat[(n - 1) * proof_len:]
    b.upload_launch(flat2, proof_len, inst, [8], rand, with_pairing=True)
    ok2, st2, _, _ = b.finish_groups()
    assert ok2 == [True] * G and st2 == [0] * n
    b.close()
    ctx.close()
