import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
import halo2_verifier_amd as h2v
d = bench.load_or_make_proofs(1024, 14, print)
ctx = h2v.Context(h2v.ParamsKZG(d["params"], h2v.SerdeFormat.RawBytes), h2v.VerifyingKey(d["vk"], h2v.SerdeFormat.RawBytes))
G = 20
proofs = d["proofs"] * G; inst = d["inst"] * G
tail = b"".join(((i * 0x9e3779b97f4a7c15 + 0x1234567) % (1 << 250)).to_bytes(32, "little") for i in range(1, 1024 * G + 1))
b = h2v.Batch(ctx, 1024 * G, 8, groups=G)
mode = sys.argv[1]
if len(sys.argv) > 2:
    ctx.set_tuning(upload_mode=int(sys.argv[2]))
for _ in range(4):
    if mode == "overlap":
        b.upload_launch(proofs, 1024, inst, [8], tail)
    else:
        b.upload(proofs, 1024, inst, [8], tail); b.launch(True)
    b.finish_groups(raw_statuses=True)
