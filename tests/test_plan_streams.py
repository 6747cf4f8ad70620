"""The plan compiler's multi-stream Fr programs (csrc/vkplan.hip: Builder::emit_streams) checked on the HOST, without a GPU:
tests/cpp/plan_host.hip compiles the plan of a VK and proves, for 2, 3 and 4 instruction streams per proof,
  * race freedom — inside a barrier epoch no stream writes a slot another stream reads or writes (the streams of a proof run in
    different waves of a workgroup; only OP_BARRIER orders them), equal barrier counts in all streams;
  * equivalence — by symbolic evaluation every STORE writes the same expression to the same place as the single-stream program.
Circuits: the headline vector_mul VK, a lookup / shuffle-heavy VK, the shuffle circuit, both multi-open schemes and transcripts,
two circuit instances per transcript, the GWC guard variant."""
import os
import subprocess

import pytest

import circuits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "halo2_verifier_amd", "csrc")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = tmp_path_factory.mktemp("plan") / "plan_host"
    cmd = ["hipcc", "-O1", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-value", "-Wno-unused-result", "-o", str(out),
           os.path.join(ROOT, "tests", "cpp", "plan_host.hip"), os.path.join(CSRC, "vkplan.hip"), os.path.join(CSRC, "params.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.fail("hipcc failed: " + r.stderr[-2000:])
    return str(out)


def _check(exe, tmp_path, s, col_lens, guard=False):
    vk, params = tmp_path / "vk", tmp_path / "params"
    vk.write_bytes(s.vk); params.write_bytes(s.params)
    m = getattr(s, "circuit_instances", 1)
    r = subprocess.run([exe, str(vk), str(params), str(s.multiopen), str(s.transcript), str(m), "1" if guard else "0"] + [str(c) for c in col_lens * m],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.split("\n")
    assert [l.split()[0] for l in lines if l.startswith("K=")] == ["K=2", "K=3", "K=4"] and all(" ok:" in l for l in lines if l.startswith("K="))
    return r.stdout


@pytest.mark.parametrize("mo,tr", [(circuits.SHPLONK, circuits.BLAKE2B), (circuits.GWC, circuits.KECCAK256)])
def test_vector_mul_streams(exe, tmp_path, mo, tr):
    s = circuits.setup_vector_mul(8, 8).set_options(mo, tr)
    out = _check(exe, tmp_path, s, [8])
    assert "19 stores" in out or mo == circuits.GWC
    if mo == circuits.GWC:
        _check(exe, tmp_path, s, [8], guard=True)      # the guard variant stores every term's own scalar
    s.free()


def test_lookup_heavy_and_shuffle_streams(exe, tmp_path):
    s = circuits.setup_wide(8, A=12, F=6, L_=2, Sh=1, deg=5)
    _, inst = circuits.prove_wide(s)
    _check(exe, tmp_path, s, [len(c) for c in inst])
    s.free()
    s = circuits.setup_shuffle(8, 4, 32).set_options(circuits.GWC, circuits.BLAKE2B)
    _, inst = circuits.prove_shuffle(s)
    _check(exe, tmp_path, s, [len(c) for c in inst])
    s.free()


def test_two_circuit_instances_streams(exe, tmp_path):
    s = circuits.setup_vector_mul(8, 4).set_circuit_instances(2)
    _check(exe, tmp_path, s, [4])
    s.free()


def test_the_checker_sees_a_race_when_the_barriers_are_removed(exe, tmp_path):
    s = circuits.setup_vector_mul(8, 8)
    vk, params = tmp_path / "vk", tmp_path / "params"
    vk.write_bytes(s.vk); params.write_bytes(s.params)
    r = subprocess.run([exe, str(vk), str(params), "0", "0", "1", "0", "8"], capture_output=True, text=True, timeout=300, env=dict(os.environ, PLAN_HOST_DROP_BARRIERS="1"))
    assert r.returncode != 0 and "without a barrier between them" in r.stderr
    s.free()
