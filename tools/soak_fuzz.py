# extended differential fuzz (not part of the suite): more seeds of tests/test_gpu_fuzz.py's headline case, and the stream-count knobs
import os, sys, random
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_fuzz as F
bad = 0
for seed in range(100, 220):
    try:
        F.test_random_corruptions_match_the_oracle(seed)
    except AssertionError as e:
        bad += 1; print("MISMATCH seed", seed, str(e)[:200])
for K in ("2", "3"):
    os.environ["H2V_FRVM_STREAMS"] = K
    for seed in range(300, 330):
        try:
            F.test_random_corruptions_match_the_oracle(seed)
        except AssertionError as e:
            bad += 1; print("MISMATCH K", K, "seed", seed, str(e)[:200])
os.environ.pop("H2V_FRVM_STREAMS")
for fam, seed in [("wide", 41), ("wide", 42), ("shuffle", 43), ("wide_m2", 45), ("shuffle_m2", 46), ("wide", 47), ("shuffle", 48)]:
    try:
        F.test_random_corruptions_other_circuits(fam, seed)
    except AssertionError as e:
        bad += 1; print("MISMATCH", fam, seed, str(e)[:200])
print("soak done, mismatches:", bad)
