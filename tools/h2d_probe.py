import ctypes, os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
import halo2_verifier_amd as h2v
d = bench.load_or_make_proofs(1024, 14, print)
ctx = h2v.Context(h2v.ParamsKZG(d["params"], h2v.SerdeFormat.RawBytes), h2v.VerifyingKey(d["vk"], h2v.SerdeFormat.RawBytes))
G = 20
proofs = d["proofs"] * G; inst = d["inst"] * G
tail = b"".join(((i * 0x9e3779b97f4a7c15 + 0x1234567) % (1 << 250)).to_bytes(32, "little") for i in range(1, 1024 * G + 1))
b = h2v.Batch(ctx, 1024 * G, 8, groups=G)
b.upload(proofs, 1024, inst, [8], tail); b.launch(True); b.finish_groups(raw_statuses=True)
hip = ctypes.CDLL("libamdhip64.so")
def t(f, n=7):
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); f(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
print("launch+finish (resident)        %.3f ms" % t(lambda: (b.launch(True), b.finish_groups(raw_statuses=True))))
print("upload only                     %.3f ms" % t(lambda: b.upload(proofs, 1024, inst, [8], tail)))
print("upload + launch + finish        %.3f ms" % t(lambda: (b.upload(proofs, 1024, inst, [8], tail), b.launch(True), b.finish_groups(raw_statuses=True))))
print("upload_launch (returns)         %.3f ms" % t(lambda: (b.upload_launch(proofs, 1024, inst, [8], tail), b.finish_groups(raw_statuses=True))[0] if False else (b.upload_launch(proofs, 1024, inst, [8], tail), b.finish_groups(raw_statuses=True))))
for mode in (1, 2, 3):
    ctx.set_tuning(upload_mode=mode)
    print("upload_launch + finish, mode %d      %.3f ms" % (mode, t(lambda: (b.upload_launch(proofs, 1024, inst, [8], tail), b.finish_groups(raw_statuses=True)), 15)))
ctx.set_tuning()
def ul():
    t0 = time.perf_counter(); b.upload_launch(proofs, 1024, inst, [8], tail); t1 = time.perf_counter(); b.finish_groups(raw_statuses=True); t2 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3
print("upload_launch call / finish     ", [tuple(round(x, 3) for x in ul()) for _ in range(5)])
buf = ctypes.c_char_p(proofs)
def reg():
    rc = hip.hipHostRegister(buf, ctypes.c_size_t(len(proofs)), 0); assert rc == 0, rc
    rc = hip.hipHostUnregister(buf); assert rc == 0, rc
print("hipHostRegister+Unregister 20MB %.3f ms" % t(reg))
