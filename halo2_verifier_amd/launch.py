"""One process per GPU: start the N ranks of a single-node job without torchrun.

`bench.py --gpus N` (and any other entry point that shards a batch) calls spawn_ranks() when it was
started as a plain process (no WORLD_SIZE in the environment).  The parent makes NO GPU call and imports neither torch
nor the HIP library: every rank is a fresh child process (never a re-exec of a process that touched the GPU) with
RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR / MASTER_PORT set the way torch.distributed.run sets
them, so the rank body is the same under both launchers.  Rank 0 inherits the parent's stdout (the one JSON line);
the other ranks' stdout is sent to stderr.  If any rank fails, the rest are terminated (a rank that exits while its
peers sit in a collective would otherwise hang the job) and the parent exits with the failing rank's code.
"""
import os
import signal
import socket
import subprocess
import sys
import time


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_env(rank: int, world: int, port: int, base=None) -> dict:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    # dmabuf IPC.  Not a finding of this repository (RCCL with more than one rank has never run in its builds: the pool hands out
    # one GPU per box): the operators of the GPU pool document that its host driver supports dmabuf IPC only and that RCCL / device
    # memory sharing across processes fails with `hipIpcGetMemHandle: invalid argument` unless this is 0; they export it on every
    # box, and a job environment built here keeps it.  An explicit setting by the caller wins.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def spawn_ranks(world: int, cmd, timeout=None, poll=0.1) -> int:
    """Run `cmd` (argv list) as `world` rank processes; returns the job's exit code (0 = every rank exited 0)."""
    if world < 1:
        raise ValueError("world size must be >= 1")
    port = free_port()
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen(list(cmd), env=rank_env(r, world, port), stdout=None if r == 0 else sys.stderr))

    def stop_all(sig=signal.SIGTERM):
        for p in procs:
            if p.poll() is None:
                try:
                    p.send_signal(sig)
                except ProcessLookupError:
                    pass

    def on_signal(signum, _frame):
        stop_all(signal.SIGTERM)
        raise SystemExit(128 + signum)

    old = {s: signal.signal(s, on_signal) for s in (signal.SIGTERM, signal.SIGINT)}
    t0 = time.monotonic()
    rc = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            failed = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if failed:
                r, c = failed[0]
                print(f"[launch] rank {r} exited with code {c}; stopping the other ranks", file=sys.stderr, flush=True)
                rc = c if c > 0 else 1
                break
            if all(c == 0 for c in codes):
                break
            if timeout is not None and time.monotonic() - t0 > timeout:
                print(f"[launch] timeout after {timeout}s; stopping all ranks", file=sys.stderr, flush=True)
                rc = 124
                break
            time.sleep(poll)
    finally:
        if any(p.poll() is None for p in procs):
            stop_all(signal.SIGTERM)
            t1 = time.monotonic()
            while any(p.poll() is None for p in procs) and time.monotonic() - t1 < 10:
                time.sleep(0.05)
            stop_all(signal.SIGKILL)
            for p in procs:
                p.wait()
        for s, h in old.items():
            signal.signal(s, h)
    return rc
