"""Latency of verifying ONE proof (SingleStrategy, kzg/strategy.rs:143-181) on the GPU: the one-shot call from host bytes, a resident
batch of one, and the stages of that launch:  python tools/single_proof_probe.py"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
bench.hw_queue_env()
import halo2_verifier_amd as h2v

d = bench.load_or_make_proofs(1024, 14, print)
ctx = h2v.Context(h2v.ParamsKZG(d["params"], h2v.SerdeFormat.RawBytes), h2v.VerifyingKey(d["vk"], h2v.SerdeFormat.RawBytes))
N = bench.N_PUBLIC
P = [d["proofs"][i * 1024:(i + 1) * 1024] for i in range(16)]
I = [[[d["inst"][(i * N + j) * 32:(i * N + j + 1) * 32] for j in range(N)]] for i in range(16)]


def med(f, n=15):
    v = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); v.append((time.perf_counter() - t0) * 1e3)
    v.sort()
    return v[0], v[len(v) // 2]


ctx.verify_each(P[:1], I[:1])
print("h2v_verify_each, 1 proof, from host bytes:   best %.3f ms, median %.3f ms" % med(lambda: ctx.verify_each(P[:1], I[:1])))
print("h2v_verify_batch, 1 proof, from host bytes:  best %.3f ms, median %.3f ms" % med(lambda: ctx.verify_batch(P[:1], I[:1], [1])))
print("h2v_verify_each, 16 proofs, from host bytes: best %.3f ms, median %.3f ms" % med(lambda: ctx.verify_each(P, I)))
b = h2v.Batch(ctx, 1, N, groups=1)
b.upload(P[0], 1024, b"".join(I[0][0]), [N], (1).to_bytes(32, "little"))
b.launch(True); b.finish_groups()
print("resident batch of one, launch + finish:      best %.3f ms, median %.3f ms" % med(lambda: (b.launch(True), b.finish_groups())))
b.set_profiling(True)
acc = {}
for _ in range(9):
    b.launch(True); b.finish_groups()
    for k, v in b.timings_ms().items():
        acc.setdefault(k, []).append(v)
print("stages of that launch (median, ms):", {k: round(sorted(v)[len(v) // 2], 3) for k, v in acc.items()})
b.close(); ctx.close()
