# extended differential fuzz (not part of the suite): more seeds of tests/test_gpu_fuzz.py's headline case, 
import os, sys, random
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_fuzz as F
bad = 0
for seed in range(100, 220):
    try:
        F.test_random_corruptions_match_the_oracle(seed)
    except AssertionError as e:
        bad += 1; print("MISMATCH seed", seed, str(e)[:200])
# (the 2- and 3-stream Fr programs, forced with h2v_ctx_set_tuning, are in the suite since round 3: tests/test_gpu_split_pairing.py)
for fam, seed in [("wide", 41), ("wide", 42), ("shuffle", 43), ("wide_m2", 45), ("shuffle_m2", 46), ("wide", 47), ("shuffle", 48)]:
    try:
        F.test_random_corruptions_other_circuits(fam, seed)
    except AssertionError as e:
        bad += 1; print("MISMATCH", fam, seed, str(e)[:200])
# the group arithmetic on its own: random MSM shapes against the oracle — sizes around the sort / cut thresholds, repeated and opposite bases
# (the additions' degenerate cases), identity bases, zero / tiny / full-width scalars
import oracle_lib, srs_util
import halo2_verifier_amd as h2v
from srs_util import R_MOD, g1_xy
P_MOD = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
oracle = oracle_lib.load()
srs = srs_util.load_srs(os.path.join("tests", "golden", "kzg_bn254_8.srs"))
ctx = h2v.Context(h2v.ParamsKZG(srs.params_raw, h2v.SerdeFormat.RawBytes))
pts = [g1_xy(p) for p in srs.g]
neg = lambda b: b[:32] + ((P_MOD - int.from_bytes(b[32:], "little")) % P_MOD).to_bytes(32, "little")
rnd = random.Random(99)
for it in range(int(os.environ.get("SOAK_MSM", "60"))):
    n = rnd.choice([rnd.randrange(1, 64), rnd.randrange(64, 3000), rnd.randrange(3000, 20000)])
    pool = rnd.choice([1, 2, 5, 50, len(pts)])
    kind = rnd.choice(["full", "full", "tiny", "few", "mixed"])
    few = [rnd.randrange(R_MOD) for _ in range(3)]
    scalars, bases = [], []
    for i in range(n):
        k = {"full": rnd.randrange(R_MOD), "tiny": rnd.randrange(300), "few": rnd.choice(few), "mixed": rnd.choice([0, 1, R_MOD - 1, rnd.randrange(R_MOD), rnd.randrange(1 << 64)])}[kind]
        b = pts[rnd.randrange(pool)]
        r = rnd.random()
        if r < 0.1: b = neg(b)
        elif r < 0.12: b = bytes(64)
        scalars.append(k); bases.append(b)
    if ctx.msm_g1(scalars, bases) != oracle_lib.g1_msm(oracle, scalars, bases):
        bad += 1; print("MSM MISMATCH", it, n, pool, kind)
# pairing verdicts (round 3: the kernels drop the final exponentiation's inversion and test membership in Fq*): SOAK_PAIR seeded cases —
# relations that hold, near misses, identities on either side, equal and opposite points — against the oracle's textbook check
import ctypes
P_ORDER_CASES = int(os.environ.get("SOAK_PAIR", "600"))
rnd = random.Random(2026)
def oracle_check(l, r):
    ok = ctypes.c_int(-1)
    assert oracle.h2o_pairing_check(srs.params_raw, len(srs.params_raw), 1, l, r, ctypes.byref(ok)) == 0
    return bool(ok.value)
pair_bad = 0
for it in range(P_ORDER_CASES):
    m = rnd.randrange(1, 4)
    idx = [rnd.randrange(0, 250) for _ in range(m)]
    a = [rnd.randrange(1, R_MOD) for _ in range(m)]
    left = oracle_lib.g1_msm(oracle, a, [pts[i] for i in idx])
    right = oracle_lib.g1_msm(oracle, a, [pts[i + 1] for i in idx])
    kind = it % 6
    if kind == 1: right = oracle_lib.g1_msm(oracle, [(a[0] + 1) % R_MOD or 1] + a[1:], [pts[i + 1] for i in idx])
    elif kind == 2: left, right = right, left
    elif kind == 3: right = neg(right)
    elif kind == 4: left = bytes(64) if it % 12 == 4 else left; right = bytes(64)
    elif kind == 5: right = left
    want, got = oracle_check(left, right), ctx.pairing_check(left, right)
    if want != got:
        pair_bad += 1; print("PAIRING MISMATCH case", it, kind, want, got)
print("pairing verdicts:", P_ORDER_CASES, "cases, mismatches:", pair_bad)
bad += pair_bad
ctx.close()
print("soak done, mismatches:", bad)
