"""Pin the oracle's Blake2b transcript against hashlib (the reference's hash dependency blake2b_simd is not
vendored): BLAKE2b-512 with personal "Halo2-Transcript" (transcript/mod.rs:126-129), the absorb prefixes
(:216-231), squeeze = absorb 0x00 + finalize a clone (:209-214), from_uniform_bytes = 512-bit LE mod r (:500-514)."""
import ctypes
import hashlib
import random

from srs_util import R_MOD


def test_blake2b_matches_hashlib(oracle):
    rnd = random.Random(7)
    for n in (0, 1, 63, 64, 127, 128, 129, 255, 256, 257, 1000, 4096):
        data = bytes(rnd.randrange(256) for _ in range(n))
        out = ctypes.create_string_buffer(64)
        oracle.h2o_blake2b_personal(b"Halo2-Transcript", data, n, out)
        assert out.raw == hashlib.blake2b(data, digest_size=64, person=b"Halo2-Transcript").digest(), n


def test_from_uniform_bytes(oracle):
    rnd = random.Random(8)
    cases = [bytes(64), b"\xff" * 64, (R_MOD).to_bytes(64, "little"), (R_MOD - 1).to_bytes(64, "little"), ((1 << 256)).to_bytes(64, "little")]
    cases += [bytes(rnd.randrange(256) for _ in range(64)) for _ in range(50)]
    for c in cases:
        out = ctypes.create_string_buffer(32)
        oracle.h2o_fr_from_uniform(c, out)
        assert int.from_bytes(out.raw, "little") == int.from_bytes(c, "little") % R_MOD


def test_challenges_of_a_real_proof_follow_the_python_restatement(oracle):
    """Replay the transcript of a vector_mul proof with hashlib and compare all 8 challenges with the oracle's trace."""
    import circuits
    s = circuits.setup_vector_mul(8, 10)
    proof, inst = circuits.prove_vector_mul(s, [2] * 10, [3] * 10)
    rc, g = circuits.oracle_guard(s, proof, inst)
    assert rc == 0
    P = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47

    def decompress(b):
        x = int.from_bytes(b, "little") & ((1 << 254) - 1)
        sign = (b[31] >> 6) & 1
        y = pow((x * x * x + 3) % P, (P + 1) // 4, P)
        assert y * y % P == (x * x * x + 3) % P
        if (y & 1) != sign:
            y = P - y
        return x, y

    h = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")
    pos = 0

    def point():
        nonlocal pos
        x, y = decompress(proof[pos:pos + 32]); pos += 32
        h.update(b"\x01" + x.to_bytes(32, "little") + y.to_bytes(32, "little"))

    def scalar():
        nonlocal pos
        h.update(b"\x02" + proof[pos:pos + 32]); pos += 32

    def squeeze():
        h.update(b"\x00")
        return int.from_bytes(h.copy().digest(), "little") % R_MOD

    # vk.transcript_repr is the last 32 bytes of the RawBytes VK (Montgomery limbs)
    repr_mont = int.from_bytes(s.vk[-32:], "little")
    repr_val = repr_mont * pow(1 << 256, -1, R_MOD) % R_MOD
    h.update(b"\x02" + repr_val.to_bytes(32, "little"))
    for v in inst[0]:
        h.update(b"\x02" + v)
    for _ in range(3): point()          # advice
    theta = squeeze()
    beta = squeeze(); gamma = squeeze()
    for _ in range(4): point()          # permutation products (P=4, chunk=1)
    point()                             # random poly
    y = squeeze()
    for _ in range(2): point()          # h pieces
    x = squeeze()
    for _ in range(20): scalar()
    sy = squeeze(); sv = squeeze()
    point()
    su = squeeze()
    point()
    assert pos == 1024
    got = [int.from_bytes(c, "little") for c in g["challenges"]]
    assert got == [theta, beta, gamma, y, x, sy, sv, su]
    s.free()


def test_keccak256_matches_python_restatement_and_known_vectors(oracle):
    """Legacy Keccak-256 (sha3 0.9.1's Keccak256, not SHA3-256): known answers + the independent Python implementation in oracle/pyref.py."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pyref

    def k(data):
        out = ctypes.create_string_buffer(32)
        oracle.h2o_keccak256(data, len(data), out)
        return out.raw

    assert k(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert k(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    assert k(b"").hex() != hashlib.sha3_256(b"").hexdigest()
    rnd = random.Random(9)
    for n in (1, 55, 134, 135, 136, 137, 271, 272, 273, 1000):
        data = bytes(rnd.randrange(256) for _ in range(n))
        assert k(data) == pyref.keccak256(data), n
