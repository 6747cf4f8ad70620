"""`python bench.py --gpus N` must really be an N-rank job (the flag was parsed and never read).

Non-GPU: the launcher (halo2_verifier_amd/launch.py) starts N fresh child processes with the torchrun environment, relays
rank 0's stdout, and brings the whole job down when one rank fails; bench.py refuses a WORLD_SIZE that differs from --gpus;
`--dry-run` proves the N-rank rendezvous over gloo without a GPU; without a GPU the real run fails loudly in every rank."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_spawn_ranks_gives_every_child_the_torchrun_environment(tmp_path):
    from halo2_verifier_amd.launch import spawn_ranks
    stub = tmp_path / "stub.py"
    stub.write_text(
        "import json, os, sys\n"
        "keys = ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'HSA_ENABLE_IPC_MODE_LEGACY')\n"
        "open(os.path.join(sys.argv[1], 'rank%s.json' % os.environ['RANK']), 'w').write(json.dumps({k: os.environ.get(k) for k in keys} | {'pid': os.getpid(), 'ppid': os.getppid()}))\n")
    rc = spawn_ranks(3, [sys.executable, str(stub), str(tmp_path)])
    assert rc == 0
    seen = [json.loads((tmp_path / f"rank{r}.json").read_text()) for r in range(3)]
    assert [s["RANK"] for s in seen] == ["0", "1", "2"] and [s["LOCAL_RANK"] for s in seen] == ["0", "1", "2"]
    assert all(s["WORLD_SIZE"] == "3" and s["LOCAL_WORLD_SIZE"] == "3" and s["MASTER_ADDR"] == "127.0.0.1" for s in seen)
    assert len({s["MASTER_PORT"] for s in seen}) == 1 and len({s["pid"] for s in seen}) == 3
    assert all(s["ppid"] == os.getpid() for s in seen)          # children of the launcher: fresh processes, not re-execs
    assert all(s["HSA_ENABLE_IPC_MODE_LEGACY"] is not None for s in seen)


def test_a_failing_rank_ends_the_job(tmp_path):
    """A rank that exits while its peers wait (in a collective, here a sleep) must not leave the job hanging."""
    from halo2_verifier_amd.launch import spawn_ranks
    stub = tmp_path / "stub.py"
    stub.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(3)\ntime.sleep(120)\n")
    t0 = time.time()
    rc = spawn_ranks(2, [sys.executable, str(stub)])
    assert rc == 3 and time.time() - t0 < 30


def test_bench_dry_run_is_an_n_rank_job():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-run"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # only rank 0 speaks on stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["steps"] == 20
    assert out["steps_per_launch"] * out["launches"] + out["remainder_steps"] == 20     # EXACTLY --steps steps are timed
    assert "rank 1/2" in r.stderr and "shard [1024, 2048)" in r.stderr


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=_clean_env(WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr and not r.stdout.strip()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--dry-run"], env=_clean_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0


def test_bench_without_a_gpu_fails_loudly_in_every_rank():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "0"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stderr.count("no CPU fallback") >= 1 and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_launch_shape_times_exactly_the_requested_steps():
    sys.path.insert(0, ROOT)
    import bench
    for steps in (1, 2, 5, 20, 31, 64, 100, 2048):
        for groups, depth in ((0, 0), (32, 8), (20, 1), (3, 0), (0, 2)):
            G, launches, rem, d = bench.launch_shape(steps, groups, depth)
            assert G * launches + rem == steps and 1 <= d <= max(launches, 1) and 0 <= rem < G


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_through_the_launcher():
    """The whole N = 2 control flow of bench.py (shards, draw tails, accumulator records, all-gather, fold, one pairing per step,
    max-over-ranks timing, the config-3 leg) started by `python bench.py --gpus 2` itself — both ranks on cuda:0, the collective
    staged through gloo because RCCL refuses two ranks on one device."""
    env = _clean_env(H2V_BENCH_BACKEND="gloo", H2V_BENCH_ONE_DEVICE="1")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "6", "--warmup", "2", "--config3-steps", "2", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["value"] > 0 and out["scaling"] == "weak"
    assert out["config3"]["proofs_per_step"] == 2 * 8192 and out["config3"]["value"] > 0
