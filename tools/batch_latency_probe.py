"""End-to-end latency of ONE batch by size (resident inputs, launch + finish, median of 11) and its MSM / accumulate stage:
   python tools/batch_latency_probe.py"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
bench.hw_queue_env()
import halo2_verifier_amd as h2v

d = bench.load_or_make_proofs(1024, 14, print)
ctx = h2v.Context(h2v.ParamsKZG(d["params"], h2v.SerdeFormat.RawBytes), h2v.VerifyingKey(d["vk"], h2v.SerdeFormat.RawBytes))
N = bench.N_PUBLIC
for n in (1, 4, 16, 64, 256, 512, 1024, 2048, 4096):
    reps = (n + 1023) // 1024
    proofs = (d["proofs"] * reps)[: n * 1024]; inst = (d["inst"] * reps)[: n * 32 * N]
    tail = b"".join(((i * 0x9e3779b97f4a7c15 + 7) % (1 << 250)).to_bytes(32, "little") for i in range(1, n + 1))
    b = h2v.Batch(ctx, n, N, groups=1)
    b.upload(proofs, 1024, inst, [N], tail)
    b.set_profiling(True)
    ts, st = [], {}
    for i in range(13):
        t0 = time.perf_counter(); b.launch(True); ok, _, _, _ = b.finish_groups(raw_statuses=True); ts.append((time.perf_counter() - t0) * 1e3)
        assert all(ok)
        for k, v in b.timings_ms().items(): st.setdefault(k, []).append(v)
    ts = sorted(ts[2:])
    med = lambda v: sorted(v[2:])[len(v[2:]) // 2]
    print("n = %4d: %.3f ms   msm %.3f (accumulate %.3f)  pairing %.3f  decompress %.3f  fr %.3f" % (n, ts[len(ts) // 2], med(st["msm"]), med(st["msm_accumulate"]), med(st["pairing"]), med(st["decompress"]), med(st["fr_program"])), flush=True)
    b.close()
ctx.close()
