"""Host cost of h2v_batch_launch and what splitting the driver's launch (20 batches of 1024 proofs) into d launches in flight gives:
   python tools/launch_probe.py
Prints, per split d in (1, 2, 4): the host time of one launch call, and the wall time from the first launch call to the last finish
when the d launches are enqueued (a) by one thread, one after the other, (b) by d threads (ctypes drops the GIL inside the call)."""
import os, sys, threading, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
bench.hw_queue_env()          # GPU_MAX_HW_QUEUES = 16 before the runtime starts (bench.py does the same)
import halo2_verifier_amd as h2v

d = bench.load_or_make_proofs(1024, 14, print)
ctx = h2v.Context(h2v.ParamsKZG(d["params"], h2v.SerdeFormat.RawBytes), h2v.VerifyingKey(d["vk"], h2v.SerdeFormat.RawBytes))
TOTAL = 20


def make(G):
    proofs = d["proofs"] * G; inst = d["inst"] * G
    tail = b"".join(((i * 0x9e3779b97f4a7c15 + 0x1234567) % (1 << 250)).to_bytes(32, "little") for i in range(1, 1024 * G + 1))
    b = h2v.Batch(ctx, 1024 * G, 8, groups=G)
    b.upload(proofs, 1024, inst, [8], tail); b.launch(True); b.finish_groups(raw_statuses=True)
    return b


def best(f, n=9):
    v = sorted(f() for _ in range(n))
    return v[0], v[len(v) // 2]


for split in (1, 2, 4):
    G = TOTAL // split
    bs = [make(G) for _ in range(split)]

    def serial():
        t0 = time.perf_counter()
        for b in bs: b.launch(True)
        t1 = time.perf_counter()
        for b in bs: b.finish_groups(raw_statuses=True)
        return (time.perf_counter() - t0) * 1e3, (t1 - t0) * 1e3

    def call_only():
        r = serial()
        return r[1] / split

    def total_serial():
        return serial()[0]

    go = [threading.Event() for _ in bs]; done = [threading.Event() for _ in bs]; stop = False

    def worker(i):
        while True:
            go[i].wait(); go[i].clear()
            if stop: return
            bs[i].launch(True)
            done[i].set()
    th = [threading.Thread(target=worker, args=(i,), daemon=True) for i in range(split)]
    for t in th: t.start()

    def threaded():
        t0 = time.perf_counter()
        for e in go: e.set()
        for e in done: e.wait(); e.clear()
        for b in bs: b.finish_groups(raw_statuses=True)
        return (time.perf_counter() - t0) * 1e3
    print("split %d x %2d groups: launch call %.3f / %.3f ms (best / median);  one thread %.3f / %.3f ms;  %d threads %.3f / %.3f ms"
          % ((split, G) + best(call_only) + best(total_serial) + (split,) + best(threaded)), flush=True)
    stop = True
    for e in go: e.set()
    for b in bs: b.close()
ctx.close()
