"""The C ABI from compiled code: tests/cpp/harness.cpp (C++17, include/h2v.hpp — the host-side mirror of the reference's
verify_proof / ParamsKZG / VerifyingKey / SingleStrategy / AccumulatorStrategy) is built with g++ against libh2v_amd.so and
run on seeded inputs that include a wrong public input and an undecodable point; every line it prints is compared with the
CPU oracle.  This is the shape of the reference's own tests (halo2_verifier/tests/helpers.rs:66-85) in the language a
maintainer without the Rust shim would call the library from."""
import os
import random
import subprocess

import pytest

import circuits
from circuits import R_MOD

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_harness_matches_oracle(tmp_path):
    from halo2_verifier_amd import _lib
    lib = _lib.lib_path()
    exe = tmp_path / "harness"
    subprocess.run(["g++", "-std=c++17", "-O1", "-o", str(exe), os.path.join(ROOT, "tests", "cpp", "harness.cpp"), lib,
                    "-Wl,-rpath," + os.path.dirname(lib)], check=True)
    s = circuits.setup_vector_mul(8, 8)
    n = 6
    P, I = circuits.prove_vector_mul_batch(s, n, seed=21, threads=4)
    P, I = list(P), list(I)
    I[2] = [[circuits.le32(77)] + I[2][0][1:]]                       # wrong public input: pairing fails
    bad = bytearray(P[4]); bad[31] = 0xff; P[4] = bytes(bad)          # first advice commitment undecodable: Error::Transcript
    rnd = random.Random(8)
    rand = [rnd.randrange(1, R_MOD) for _ in range(n)]
    d = tmp_path
    (d / "params.bin").write_bytes(s.params)
    (d / "vk.bin").write_bytes(s.vk)
    (d / "proofs.bin").write_bytes(b"".join(P))
    (d / "inst.bin").write_bytes(b"".join(b"".join(col) for i in I for col in i))
    (d / "rand.bin").write_bytes(b"".join(r.to_bytes(32, "little") for r in rand))
    (d / "meta.txt").write_text(f"{n} {len(P[0])} 8\n")
    out = subprocess.run([str(exe), str(d)], check=True, capture_output=True, text=True, timeout=300).stdout.splitlines()
    singles = [int(l.split()[2]) for l in out if l.startswith("single ")]
    assert singles == [circuits.oracle_verify_single(s, p, i) for p, i in zip(P, I)] == [0, 0, -2, 0, -5, 0]
    assert [l for l in out if l.startswith("single_bad_columns")] == ["single_bad_columns -1"]
    b = [l for l in out if l.startswith("batch ")][0].split()
    ok, st, left, right = circuits.oracle_verify_batch(s, P, I, rand)
    assert (b[1] == "1") == ok and bytes.fromhex(b[2]) == left and bytes.fromhex(b[3]) == right and [int(x) for x in b[4:]] == st
    assert ok is False
    # AccumulatorStrategy::with: the accumulation resumed from its first half gives the same two points (the verdict differs only
    # through proof 2's failing pairing, which both see)
    r = [l for l in out if l.startswith("resumed ")][0].split()
    assert bytes.fromhex(r[2]) == left and bytes.fromhex(r[3]) == right and r[1] == "0"
    s.free()
