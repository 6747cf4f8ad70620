"""The pairing over SPLIT accumulators (csrc/msm.hip: msm_final_parts, csrc/pairing.hip: k_pair_lines).

A launch that ends in its own pairing checks leaves every accumulator as `parts` points,
    acc = sum_j 2^(shift j) piece_j,
and checks  prod_j e(L_j, 2^(shift j) s_g2) e(R_j, -2^(shift j) g2) == 1  against tables of G2 multiples built on the host,
while the whole points are put together beside the pairing.  Whatever the number of pieces (h2v_tuning.msm_parts, read per launch;
1 = the unsplit path), verdicts, per-proof statuses and the evaluated accumulator channels must be the oracle's, bit for
bit — for accepted batches, for a batch whose pairing fails, with a rejected proof inside, for grouped launches, for one proof
per check (SingleStrategy) and for both multi-open schemes (different window plans, i.e. different shifts)."""
import random

import pytest

import circuits
from circuits import R_MOD

pytestmark = pytest.mark.gpu

PARTS = [1, 2, 3, 4, 5, 6]


class _tuned:
    """Forced kernel variants (h2v_ctx_set_tuning) for the launches inside the block; automatic choice again afterwards."""

    def __init__(self, ctx, **fields):
        self.ctx, self.fields = ctx, fields

    def __enter__(self):
        self.ctx.set_tuning(**self.fields)

    def __exit__(self, *a):
        self.ctx.set_tuning()


@pytest.fixture(scope="module")
def pool():
    s = circuits.setup_vector_mul(8, 8)
    P, I = circuits.prove_vector_mul_batch(s, 96, seed=777, threads=16)
    yield s, P, I
    s.free()


def _ctx(s):
    import halo2_verifier_amd as h2v
    return h2v.Context(h2v.ParamsKZG(s.params, h2v.SerdeFormat.RawBytes), h2v.VerifyingKey(s.vk, h2v.SerdeFormat.RawBytes),
                       multiopen=s.multiopen, transcript=s.transcript)


def test_every_part_count_matches_the_oracle(pool):
    s, P, I = pool
    ctx = _ctx(s)
    rnd = random.Random(5)
    for n in (1, 2, 33, 96):
        rand = [rnd.randrange(1, R_MOD) for _ in range(n)]
        exp = circuits.oracle_verify_batch(s, P[:n], I[:n], rand)
        assert exp[0] is True
        for parts in PARTS:
            with _tuned(ctx, msm_parts=parts):
                assert ctx.verify_batch(P[:n], I[:n], rand) == exp, (n, parts)
    ctx.close()


def test_failing_pairing_and_rejected_proof(pool):
    s, P, I = pool
    ctx = _ctx(s)
    n = 24
    rnd = random.Random(6)
    rand = [rnd.randrange(1, R_MOD) for _ in range(n)]
    # a wrong public input: every proof well-formed, the pairing fails (tests/vector_mul.rs:329-330)
    I_bad = list(I[:n]); I_bad[7] = [[circuits.le32(5)] + I[7][0][1:]]
    exp_bad = circuits.oracle_verify_batch(s, P[:n], I_bad, rand)
    assert exp_bad[0] is False and exp_bad[1] == [0] * n
    # an undecodable opening point: that proof is rejected and leaves the accumulators, the pairing over the rest passes
    P_rej = list(P[:n]); b = bytearray(P_rej[11]); b[-33] = 0xff; P_rej[11] = bytes(b)
    exp_rej = circuits.oracle_verify_batch(s, P_rej, I[:n], rand)
    assert exp_rej[0] is False and exp_rej[1][11] != 0
    for parts in PARTS:
        with _tuned(ctx, msm_parts=parts):
            assert ctx.verify_batch(P[:n], I_bad, rand) == exp_bad, parts
            assert ctx.verify_batch(P_rej, I[:n], rand) == exp_rej, parts
    ctx.close()


def test_grouped_launch_and_single_strategy(pool):
    import halo2_verifier_amd as h2v
    s, P, I = pool
    ctx = _ctx(s)
    G, gs = 6, 16
    n = G * gs
    rnd = random.Random(7)
    rand = [rnd.randrange(1, R_MOD) for _ in range(n)]
    I2 = list(I[:n]); I2[2 * gs + 1] = [[circuits.le32(9)] + I[2 * gs + 1][0][1:]]      # group 2's pairing fails
    flat, inst = b"".join(P[:n]), b"".join(b"".join(col) for i in I2 for col in i)
    results = []
    for parts in (1, 3, 6):
        with _tuned(ctx, msm_parts=parts):
            b = h2v.Batch(ctx, n, 8, groups=G)
            b.upload(flat, 1024, inst, [8], b"".join(r.to_bytes(32, "little") for r in rand))
            b.launch(with_pairing=True)
            results.append(b.finish_groups())
            raw = b.finish_groups(raw_statuses=True)      # the same call with the statuses as the C array's bytes
            assert raw[0] == results[-1][0] and raw[2:] == results[-1][2:]
            assert raw[1] == b"".join(v.to_bytes(4, "little", signed=True) for v in results[-1][1])
            b.close()
            # one check per proof: every proof its own accumulator pair, its own pieces
            assert ctx.verify_each(P[:20], I2[:20]) == [0] * 20
            assert ctx.verify_each(P[2 * gs:2 * gs + 3], I2[2 * gs:2 * gs + 3]) == [0, -2, 0]
    assert results[0] == results[1] == results[2]
    ok, st, left, right = results[0]
    assert ok == [True, True, False, True, True, True] and st == [0] * n
    for g in (0, 2, 5):
        sl = slice(g * gs, (g + 1) * gs)
        assert circuits.oracle_verify_batch(s, P[sl], I2[sl], rand[sl]) == (ok[g], st[sl], left[g], right[g])
    ctx.close()


def test_gwc_plan_has_its_own_shift():
    """GWC has more left-channel terms and a different largest problem, hence another window plan: other G2 multiples."""
    s = circuits.setup_vector_mul(8, 7).set_options(circuits.GWC, circuits.KECCAK256)
    P, I = circuits.prove_vector_mul_batch(s, 12, seed=31, threads=8)
    ctx = _ctx(s)
    rand = [random.Random(8).randrange(1, R_MOD) for _ in range(12)]
    exp = circuits.oracle_verify_batch(s, P, I, rand)
    assert exp[0] is True
    for parts in (1, 2, 6):
        with _tuned(ctx, msm_parts=parts):
            assert ctx.verify_batch(P, I, rand) == exp, parts
    ctx.close()
    s.free()


@pytest.mark.parametrize("knobs", [dict(pairing_one_stream=1), dict(frvm_streams=1), dict(pairing_one_stream=1, frvm_streams=1), dict(frvm_streams=2), dict(frvm_streams=3),
                                   dict(frvm_streams=2, frvm_lds_kb=36), dict(frvm_streams=4, frvm_lds_kb=78), dict(msm_global_sort=1), dict(msm_parts=1, frvm_streams=1),
                                   dict(msm_window_threads=64, msm_window_slots=3), dict(msm_window_threads=64, msm_window_wpw=2), dict(msm_window_threads=128, msm_window_wpw=2),
                                   dict(msm_window_threads=256), dict(msm_acc_waves=4), dict(msm_acc_waves=4, msm_global_sort=1)])
def test_single_stream_and_fallback_kernels_stay_exact(pool, knobs):
    """The default path runs the Fr program and the pairing as two instruction streams each and sorts inside LDS; the single-stream
    interpreter, the single-stream pairing table over merged lines, the whole-point pairing and the global counting sort remain in
    the library (other launch shapes; forced here through h2v_ctx_set_tuning) — and so do the Fr program with two and three streams and small
    LDS slices (what a throughput launch runs) and the window reduction's other shapes — same verdicts, statuses and accumulator bytes as the oracle through every one."""
    s, P, I = pool
    n = 40
    rnd = random.Random(11)
    rand = [rnd.randrange(1, R_MOD) for _ in range(n)]
    I_bad = list(I[:n]); I_bad[3] = [[circuits.le32(8)] + I[3][0][1:]]
    P_rej = list(P[:n]); b = bytearray(P_rej[17]); b[40] ^= 0x55; P_rej[17] = bytes(b)
    exp = circuits.oracle_verify_batch(s, P[:n], I[:n], rand)
    exp_bad = circuits.oracle_verify_batch(s, P[:n], I_bad, rand)
    exp_rej = circuits.oracle_verify_batch(s, P_rej, I[:n], rand)
    assert exp[0] is True and exp_bad[0] is False
    ctx = _ctx(s)
    with _tuned(ctx, **knobs):
        assert ctx.verify_batch(P[:n], I[:n], rand) == exp
        assert ctx.verify_batch(P[:n], I_bad, rand) == exp_bad
        assert ctx.verify_batch(P_rej, I[:n], rand) == exp_rej
        assert ctx.verify_each(P[:6], I_bad[:6]) == [0, 0, 0, -2, 0, 0]
    ctx.close()


@pytest.mark.parametrize("wpw", [2, 4])
def test_several_windows_per_workgroup_on_small_batches(pool, wpw):
    """Small batches have narrow windows (16 .. 64 buckets, one lane per bucket): with several windows per workgroup forced, a wave
    holds lanes of several windows and the window reduction's butterfly runs over 16- or 32-lane segments (csrc/msm.hip: msm_window,
    both its one-wave form and the form with several waves).  Same accumulators as the oracle."""
    s, P, I = pool
    ctx = _ctx(s)
    rnd = random.Random(31 + wpw)
    for n in (1, 3, 10, 20, 64):
        rand = [rnd.randrange(1, R_MOD) for _ in range(n)]
        exp = circuits.oracle_verify_batch(s, P[:n], I[:n], rand)
        with _tuned(ctx, msm_window_wpw=wpw):
            assert ctx.verify_batch(P[:n], I[:n], rand) == exp, n
    ctx.close()
