"""GPU parity for the two group-level entry points of the C ABI, against the reference's own
known answers (tests/golden/kzg_bn254_8.srs) and against the CPU oracle on seeded inputs."""
import ctypes
import random

import pytest

import oracle_lib
import srs_util
from srs_util import R_MOD, g1_xy

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(srs):
    import halo2_verifier_amd as h2v
    c = h2v.Context(h2v.ParamsKZG(srs.params_raw, h2v.SerdeFormat.RawBytes))
    yield c
    c.close()


def test_msm_reproduces_lagrange_basis(ctx, srs):
    n = srs.n
    w_inv = pow(srs_util.omega(srs.k), -1, R_MOD)
    n_inv = pow(n, -1, R_MOD)
    bases = [g1_xy(p) for p in srs.g]
    for j in (0, 1, 5, 100, 255):
        scalars = [pow(w_inv, i * j, R_MOD) * n_inv % R_MOD for i in range(n)]
        assert ctx.msm_g1(scalars, bases) == g1_xy(srs.g_lagrange[j]), j
    assert ctx.msm_g1([1] * n, [g1_xy(p) for p in srs.g_lagrange]) == g1_xy((1, 2))


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 31, 32, 33, 100, 1000])
def test_msm_matches_oracle(ctx, srs, oracle, n):
    rnd = random.Random(1000 + n)
    scalars = [rnd.randrange(R_MOD) for _ in range(n)]
    if n > 3:
        scalars[0] = 0
        scalars[1] = R_MOD - 1
        scalars[2] = 1
    bases = [g1_xy(srs.g[rnd.randrange(srs.n)]) for _ in range(n)]
    if n > 5:
        bases[4] = bytes(64)                # identity base
        bases[5] = bases[3]                 # repeated base
    assert ctx.msm_g1(scalars, bases) == oracle_lib.g1_msm(oracle, scalars, bases)


@pytest.mark.parametrize("n", [16384, 16385, 40000])
def test_msm_large_problem_is_cut(ctx, srs, oracle, n):
    """More terms than one per-window LDS sort takes (16 384): the problem is cut into sub-problems whose window sums are merged
    (csrc/msm.hip: msm_merge_windows).  Same bytes as the oracle, and as the uncut form (global counting sort)."""
    rnd = random.Random(7000 + n)
    pts = [g1_xy(p) for p in srs.g]
    scalars = [rnd.randrange(R_MOD) for _ in range(n)]
    scalars[5] = 0
    scalars[n - 1] = R_MOD - 1
    scalars[16383] = 1
    bases = [pts[rnd.randrange(srs.n)] for _ in range(n)]
    bases[9] = bytes(64)
    exp = oracle_lib.g1_msm(oracle, scalars, bases)
    assert ctx.msm_g1(scalars, bases) == exp
    ctx.set_tuning(msm_no_term_split=1)
    try:
        assert ctx.msm_g1(scalars, bases) == exp
    finally:
        ctx.set_tuning()


def test_msm_large_skewed_problem(ctx, srs, oracle):
    """A cut problem whose sub-problems are all one bucket per window (workgroup fix-up inside every sub-problem)"""
    rnd = random.Random(8)
    n = 33000
    pts = [g1_xy(p) for p in srs.g]
    k = rnd.randrange(R_MOD)
    scalars, bases = [k] * n, [pts[rnd.randrange(srs.n)] for _ in range(n)]
    assert ctx.msm_g1(scalars, bases) == oracle_lib.g1_msm(oracle, scalars, bases)


@pytest.mark.parametrize("kind", ["one_scalar", "one_base", "same_term", "two_values", "small_scalars"])
def test_msm_skewed_inputs(ctx, srs, oracle, kind):
    """Inputs that defeat the 'random scalars' assumptions of the bucket method: one bucket per window holds every
    entry (its pieces are spread over hundreds of accumulation chunks -> the workgroup fix-up), the same point meets
    itself inside a chunk (the complete-formula fallback), and buckets whose partial sums coincide or cancel."""
    rnd = random.Random({"one_scalar": 1, "one_base": 2, "same_term": 3, "two_values": 4, "small_scalars": 5}[kind])
    n = 5000
    pts = [g1_xy(p) for p in srs.g]
    if kind == "one_scalar":        # every term in the same bucket of every window
        k = rnd.randrange(R_MOD)
        scalars, bases = [k] * n, [pts[rnd.randrange(srs.n)] for _ in range(n)]
    elif kind == "one_base":        # buckets are multiples of one point: equal and opposite partial sums everywhere
        scalars, bases = [rnd.randrange(R_MOD) for _ in range(n)], [pts[3]] * n
    elif kind == "same_term":       # P + P in every chunk
        scalars, bases = [0x1234567890abcdef1234567890abcdef] * n, [pts[9]] * n
    elif kind == "two_values":
        a, b = rnd.randrange(R_MOD), rnd.randrange(R_MOD)
        scalars = [a if i % 3 else b for i in range(n)]
        bases = [pts[i % 7] for i in range(n)]
    else:                           # all windows above the first are empty
        scalars, bases = [rnd.randrange(1, 200) for _ in range(n)], [pts[rnd.randrange(srs.n)] for _ in range(n)]
    assert ctx.msm_g1(scalars, bases) == oracle_lib.g1_msm(oracle, scalars, bases)


def test_msm_cancellation_gives_identity(ctx, srs):
    p = srs.g[7]
    q = (p[0], srs_util.P - p[1])
    assert ctx.msm_g1([5, 5], [g1_xy(p), g1_xy(q)]) == bytes(64)
    assert ctx.msm_g1([3, R_MOD - 3], [g1_xy(p), g1_xy(p)]) == bytes(64)


def test_msm_rejects_bad_inputs(ctx, srs):
    import halo2_verifier_amd as h2v
    with pytest.raises(h2v.H2VError):
        ctx.msm_g1([R_MOD], [g1_xy(srs.g[1])])          # scalar not canonical
    with pytest.raises(h2v.H2VError):
        ctx.msm_g1([1], [g1_xy((1, 3))])                # not on the curve


def test_pairing_relations(ctx, srs):
    for i in (0, 1, 100, 254):
        assert ctx.pairing_check(g1_xy(srs.g[i]), g1_xy(srs.g[i + 1])) is True
    assert ctx.pairing_check(g1_xy(srs.g[3]), g1_xy(srs.g[5])) is False
    assert ctx.pairing_check(bytes(64), bytes(64)) is True     # empty accumulator
    assert ctx.pairing_check(g1_xy(srs.g[3]), bytes(64)) is False


def test_pairing_verdicts_random_against_oracle(ctx, srs, oracle):
    """DualMSM::check verdicts on seeded pairs, GPU against the oracle's textbook final exponentiation.  The GPU kernels carry the
    scalar nu = |Norm(f conj f)|^2 through the final exponentiation instead of inverting it and test membership in Fq* at the end
    (csrc/pairing.hip: pairing_program): the same verdict for every input, checked here on relations that hold (left = sum a_i [s^i],
    right = sum a_i [s^(i+1)]) and on near misses (one coefficient off by one, sides swapped, one side the identity)."""
    rnd = random.Random(99)
    L = oracle
    cases = []
    for t in range(12):
        m = rnd.randrange(1, 6)
        idx = [rnd.randrange(0, 250) for _ in range(m)]
        a = [rnd.randrange(1, R_MOD) for _ in range(m)]
        left = oracle_lib.g1_msm(L, a, [g1_xy(srs.g[i]) for i in idx])
        right = oracle_lib.g1_msm(L, a, [g1_xy(srs.g[i + 1]) for i in idx])
        cases.append((left, right))                                   # holds
        b = list(a); b[0] = (b[0] + 1) % R_MOD or 1
        cases.append((left, oracle_lib.g1_msm(L, b, [g1_xy(srs.g[i + 1]) for i in idx])))   # one coefficient off
        cases.append((right, left))                                   # swapped
        if t % 4 == 0:
            cases.append((left, bytes(64)))
            cases.append((bytes(64), right))
    want = []
    for left, right in cases:
        ok = ctypes.c_int(-1)
        assert L.h2o_pairing_check(srs.params_raw, len(srs.params_raw), 1, left, right, ctypes.byref(ok)) == 0
        want.append(bool(ok.value))
    got = [ctx.pairing_check(left, right) for left, right in cases]
    assert got == want
    assert want.count(True) == 12 and want.count(False) == len(cases) - 12
