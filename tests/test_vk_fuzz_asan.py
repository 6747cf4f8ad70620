"""The parsers of untrusted bytes under AddressSanitizer on the CPU (GPU sanitizers are not available on the pool): VerifyingKey::read /
ParamsKZG::read_custom restated in csrc/vkplan.hip / params.hip, the writers of csrc/serde.hip and the plan compiler are fed truncated,
bit-flipped and count-inflated keys (tests/cpp/fuzz_vk.hip).  Round 3 found with it that cs_degree — the one count of the format that
consumes no bytes — was unbounded: one flipped bit made the plan compiler lay out 2^31 quotient commitments."""
import os
import subprocess

import pytest

import circuits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "halo2_verifier_amd", "csrc")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = tmp_path_factory.mktemp("fuzz") / "fuzz_vk"
    cmd = ["hipcc", "-O1", "-g", "-std=c++17", "--offload-arch=gfx950", "-fsanitize=address", "-fno-gpu-sanitize", "-Wno-unused-value", "-Wno-unused-result",
           "-Wno-comment", "-o", str(out), os.path.join(ROOT, "tests", "cpp", "fuzz_vk.hip"), os.path.join(CSRC, "vkplan.hip"), os.path.join(CSRC, "params.hip"),
           os.path.join(CSRC, "serde.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.fail("hipcc failed: " + r.stderr[-2000:])
    return str(out)


@pytest.mark.parametrize("which", ["vector_mul", "lookup_shuffle"])
def test_mutated_keys_never_read_out_of_bounds_or_run_away(exe, tmp_path, which):
    s = circuits.setup_vector_mul(8, 8) if which == "vector_mul" else circuits.setup_wide(8, A=12, F=6, L_=2, Sh=1, deg=5)
    vk, params = tmp_path / "vk", tmp_path / "params"
    vk.write_bytes(s.vk); params.write_bytes(s.params)
    s.free()
    r = subprocess.run([exe, str(vk), str(params), "1500"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    assert "fuzzed 1500 VKs" in r.stdout


def test_cs_degree_is_bounded_like_the_reference_domain():
    """EvaluationDomain::new asserts extended_k <= 28 (poly/domain.rs:44-50): a key whose cs_degree breaks that is refused"""
    import halo2_verifier_amd as h2v
    s = circuits.setup_vector_mul(8, 8)
    nfix = int.from_bytes(s.vk[4:8], "big")
    at = 8 + 64 * nfix                                     # k | n_fixed | fixed commitments (RawBytes: 64 B each) | cs_degree
    assert int.from_bytes(s.vk[at:at + 4], "big") == 3
    for deg, ok in ((3, True), ((1 << 20) + 1, True), ((1 << 20) + 2, False), (0x80000003, False)):
        bad = s.vk[:at] + deg.to_bytes(4, "big") + s.vk[at + 4:]
        try:
            h2v.VerifyingKey(bad, h2v.SerdeFormat.RawBytes).to_bytes(h2v.SerdeFormat.Processed)
            assert ok, deg
        except h2v.H2VError as e:
            assert not ok and e.code == -17, deg
    s.free()
