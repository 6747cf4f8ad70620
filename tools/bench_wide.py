#!/usr/bin/env python3
"""Ad-hoc measurements beside the headline bench: BASELINE.json config 4 (lookup-heavy VK: 32 advice, 16 fixed, 8 two-column
lookups, degree 5) and the other multi-open / transcript instantiations on the headline VK (--circuit vector_mul):
batches of `--batch` proofs, `--groups` batches per launch, `--depth` launches in flight.  Not the headline bench (bench.py); numbers quoted in DESIGN.md."""
import argparse, os, sys, time
os.environ["GPU_MAX_HW_QUEUES"] = str(min(16, int(os.environ.get("GPU_MAX_HW_QUEUES", "16") or 16)))   # clamped: see bench.py hw_queue_env()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import circuits
import halo2_verifier_amd as h2v

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--depth", type=int, default=8)
ap.add_argument("--groups", type=int, default=8)
ap.add_argument("--steps", type=int, default=64)
ap.add_argument("--distinct", type=int, default=32)
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--circuit", choices=["wide", "vector_mul"], default="wide", help="wide = the config-4 VK; vector_mul = the headline VK (for the GWC / Keccak instantiations)")
ap.add_argument("--multiopen", choices=["shplonk", "gwc"], default="shplonk")
ap.add_argument("--transcript", choices=["blake2b", "keccak256"], default="blake2b")
a = ap.parse_args()
mo = circuits.GWC if a.multiopen == "gwc" else circuits.SHPLONK
trk = circuits.KECCAK256 if a.transcript == "keccak256" else circuits.BLAKE2B
if a.circuit == "wide":
    s = circuits.setup_wide(a.k, A=32, F=16, L_=8, Sh=0, deg=5).set_options(mo, trk)
    P, I = [], []
    for i in range(a.distinct):
        p, inst = circuits.prove_wide(s, witness_seed=i, rng_seed=1000 + i)
        P.append(p); I.append(inst)
else:
    s = circuits.setup_vector_mul(a.k, 8).set_options(mo, trk)
    P, I = circuits.prove_vector_mul_batch(s, a.distinct, seed=77, threads=16)
plen = len(P[0])
reps = (a.batch + a.distinct - 1) // a.distinct
G = a.groups
pf = (b"".join(P) * reps)[: a.batch * plen] * G
inf = (b"".join(b"".join(c) for i in I for c in i) * reps)[: a.batch * 8 * 32] * G
ctx = h2v.Context(h2v.ParamsKZG(s.params, h2v.SerdeFormat.RawBytes), h2v.VerifyingKey(s.vk, h2v.SerdeFormat.RawBytes), multiopen=mo, transcript=trk)
print("shape", ctx.proof_shape())
tail = b"".join(((i * 0x9e3779b97f4a7c15 + 77) % (1 << 250)).to_bytes(32, "little") for i in range(1, a.batch * G + 1))
bs = []
for _ in range(a.depth):
    b = h2v.Batch(ctx, a.batch * G, 8, groups=G); b.upload(pf, plen, inf, [8], tail); b.set_profiling(True); bs.append(b)
fl = [False] * a.depth
def run(n):
    for st in range(n):
        i = st % a.depth
        if fl[i]:
            ok, stt, _, _ = bs[i].finish_groups(raw_statuses=True); assert all(ok) and stt.count(0) == len(stt)
        bs[i].launch(True); fl[i] = True
    for i in range(a.depth):
        if fl[i]:
            ok, stt, _, _ = bs[i].finish_groups(raw_statuses=True); assert all(ok) and stt.count(0) == len(stt); fl[i] = False
run(a.depth)
t = time.perf_counter(); run(a.steps); dt = time.perf_counter() - t
print(f"{a.circuit} VK, {a.multiopen}/{a.transcript}: batch {a.batch} x {G} groups per launch, depth {a.depth}: {a.batch * G * a.steps / dt:.0f} proofs/s, {dt / a.steps / G * 1e3:.3f} ms per {a.batch}-proof batch, proof {plen} B, stages {bs[0].timings_ms()}")
