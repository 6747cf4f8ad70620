"""AccumulatorStrategy::with(msm_accumulator) (kzg/strategy.rs:75-78) — the reference's only pause / resume hook — through
h2v_verify_batch_seeded: a batch that starts from an existing DualMSM instead of an empty one.

Every later process() scales the whole accumulator by its fresh draw (strategy.rs:129), so the seed's terms are multiplied by the
product of ALL draws of the seeded call.  Expected values are restated with the CPU oracle: M x eval(seed) + the batch's own
accumulators, one pairing."""
import random

import pytest

import circuits
import oracle_lib
from circuits import R_MOD

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pool():
    s = circuits.setup_vector_mul(8, 8)
    P, I = circuits.prove_vector_mul_batch(s, 24, seed=555, threads=8)
    yield s, P, I
    s.free()


def _ctx(s):
    import halo2_verifier_amd as h2v
    return h2v.Context(h2v.ParamsKZG(s.params, h2v.SerdeFormat.RawBytes), h2v.VerifyingKey(s.vk, h2v.SerdeFormat.RawBytes))


def test_resume_equals_one_accumulation(pool):
    """verify_batch(first part) -> (L, R); verify_batch(second part, seed = (1, L), (1, R)) == ONE verify_batch over everything with
    the draws concatenated — accumulator bytes and verdict — for several cut points, and chained three times."""
    s, P, I = pool
    ctx = _ctx(s)
    rnd = random.Random(1)
    rand = [rnd.randrange(1, R_MOD) for _ in P]
    whole = ctx.verify_batch(P, I, rand)
    assert whole[0] is True and circuits.oracle_verify_batch(s, P, I, rand) == whole
    for cut in (1, 7, 23):
        ok1, st1, L, R = ctx.verify_batch(P[:cut], I[:cut], rand[:cut])
        ok2, st2, L2, R2 = ctx.verify_batch(P[cut:], I[cut:], rand[cut:], seed=(([1], [L]), ([1], [R])))
        assert (ok1 and ok2, st1 + st2, L2, R2) == whole, cut
    # three legs
    _, _, L, R = ctx.verify_batch(P[:5], I[:5], rand[:5])
    _, _, L, R = ctx.verify_batch(P[5:11], I[5:11], rand[5:11], seed=(([1], [L]), ([1], [R])))
    got = ctx.verify_batch(P[11:], I[11:], rand[11:], seed=(([1], [L]), ([1], [R])))
    assert (got[0], got[2], got[3]) == (whole[0], whole[2], whole[3])
    ctx.close()


def test_seed_given_as_msm_terms(pool, srs):
    """The seed as the reference holds it — two (scalar, base) term lists — including an empty channel, an identity base, a zero
    scalar; a seed that does not satisfy the pairing relation makes the (otherwise valid) batch fail; a rejected proof still rejects."""
    import srs_util
    s, P, I = pool
    L_ = oracle_lib.load()
    ctx = _ctx(s)
    rnd = random.Random(2)
    n = 9
    rand = [rnd.randrange(1, R_MOD) for _ in range(n)]
    pts = [srs_util.g1_xy(p) for p in srs.g[:6]]
    M = 1
    for r in rand:
        M = M * r % R_MOD
    base = ctx.verify_batch(P[:n], I[:n], rand)
    assert base[0] is True
    # a seed that is itself a valid accumulator: the (left, right) of another accepted batch, split into several terms
    other = ctx.verify_batch(P[n:n + 6], I[n:n + 6], [rnd.randrange(1, R_MOD) for _ in range(6)])
    assert other[0] is True
    a, b = rnd.randrange(1, R_MOD), rnd.randrange(1, R_MOD)
    left_terms = ([a, (1 - a) % R_MOD, 0, 5], [other[2], other[2], pts[1], bytes(64)])       # a L + (1 - a) L + 0 P + 5 O = L
    right_terms = ([b, (1 - b) % R_MOD], [other[3], other[3]])
    got = ctx.verify_batch(P[:n], I[:n], rand, seed=(left_terms, right_terms))
    exp_left = oracle_lib.g1_msm(L_, [M, 1], [other[2], base[2]])
    exp_right = oracle_lib.g1_msm(L_, [M, 1], [other[3], base[3]])
    assert got == (True, [0] * n, exp_left, exp_right) and circuits.oracle_pairing_check(s, exp_left, exp_right)
    # a seed with an empty right channel: arbitrary points on the left only -> the pairing fails, the accumulators are still exact
    sc = [rnd.randrange(R_MOD) for _ in range(4)]
    got = ctx.verify_batch(P[:n], I[:n], rand, seed=((sc, pts[:4]), ([], [])))
    seed_left = oracle_lib.g1_msm(L_, sc, pts[:4])
    exp_left = oracle_lib.g1_msm(L_, [M, 1], [seed_left, base[2]])
    assert got == (False, [0] * n, exp_left, base[3])
    # no proofs at all: the seed alone is checked (finalize() right after with())
    assert ctx.verify_batch([], [], [], seed=(([1], [other[2]]), ([1], [other[3]]))) == (True, [], other[2], other[3])
    # a rejected proof rejects the seeded batch too
    bad = list(P[:n]); bb = bytearray(bad[4]); bb[0:32] = b"\xff" * 32; bad[4] = bytes(bb)
    got = ctx.verify_batch(bad, I[:n], rand, seed=(left_terms, right_terms))
    assert got[0] is False and got[1][4] == -5
    # malformed seeds are refused, not verified
    import halo2_verifier_amd as h2v
    with pytest.raises(h2v.H2VError):
        ctx.verify_batch(P[:n], I[:n], rand, seed=(([1], [b"\x01" * 64]), ([], [])))            # not on the curve
    with pytest.raises(h2v.H2VError):
        ctx.verify_batch(P[:n], I[:n], rand, seed=(([R_MOD], [pts[0]]), ([], [])))               # scalar not canonical
    ctx.close()


def test_strategy_mirror_with_accumulator(pool):
    import halo2_verifier_amd as h2v
    s, P, I = pool
    params, vk = h2v.ParamsKZG(s.params, h2v.SerdeFormat.RawBytes), h2v.VerifyingKey(s.vk, h2v.SerdeFormat.RawBytes)
    rnd = random.Random(3)
    rand = [rnd.randrange(1, R_MOD) for _ in range(10)]
    first = h2v.AccumulatorStrategy(params, rand=rand[:4])
    for p, i in zip(P[:4], I[:4]):
        first = h2v.verify_proof(params, vk, first, i, p)
    assert first.finalize() is True
    second = h2v.AccumulatorStrategy.with_accumulator(params, ([1], [first.left_xy]), ([1], [first.right_xy]), rand=rand[4:])
    for p, i in zip(P[4:10], I[4:10]):
        second = h2v.verify_proof(params, vk, second, i, p)
    assert second.finalize() is True
    ctx = _ctx(s)
    whole = ctx.verify_batch(P[:10], I[:10], rand)
    assert (second.left_xy, second.right_xy) == (whole[2], whole[3])
    ctx.close()


@pytest.mark.parametrize("mo,tr,m", [(circuits.GWC, circuits.KECCAK256, 1), (circuits.SHPLONK, circuits.BLAKE2B, 2)])
def test_resume_with_other_instantiations(mo, tr, m):
    """The seed is a pair of G1 accumulators: nothing in it depends on the multi-open scheme, the transcript hash or the number of
    circuit instances per transcript — GWC / Keccak-256 and two instances per transcript resume like the headline instantiation."""
    import halo2_verifier_amd as h2v
    s = circuits.setup_vector_mul(8, 6).set_options(mo, tr).set_circuit_instances(m)
    if m == 1:
        P, I = circuits.prove_vector_mul_batch(s, 9, seed=77, threads=4)
    else:
        pairs = [circuits.prove_vector_mul_multi(s, m, seed=100 + i, rng_seed=7 + i) for i in range(9)]
        P, I = [p for p, _ in pairs], [i for _, i in pairs]
    ctx = h2v.Context(h2v.ParamsKZG(s.params, h2v.SerdeFormat.RawBytes), h2v.VerifyingKey(s.vk, h2v.SerdeFormat.RawBytes), multiopen=mo, transcript=tr, circuit_instances=m)
    rnd = random.Random(17)
    rand = [rnd.randrange(1, R_MOD) for _ in P]
    whole = ctx.verify_batch(P, I, rand)
    assert whole[0] is True and whole == circuits.oracle_verify_batch(s, P, I, rand)
    ok1, st1, L, R = ctx.verify_batch(P[:4], I[:4], rand[:4])
    ok2, st2, L2, R2 = ctx.verify_batch(P[4:], I[4:], rand[4:], seed=(([1], [L]), ([1], [R])))
    assert (ok1 and ok2, st1 + st2, L2, R2) == whole
    ctx.close()
    s.free()
