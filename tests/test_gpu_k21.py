"""The reference's one real-world shape: 2^19 public inputs at k = 21 (serialize/examples/vector_mul.rs:306-310), without a k = 21
prover: the VK comes from the known-s keygen (67 s of CPU; cached under tests/golden/_cache/ when it has been made before), the
"proof" is 1024 bytes of decodable random points and canonical random scalars.  Everything up to the verdict is a deterministic
function of (VK, instances, proof bytes): the Fiat-Shamir challenges, the instance evaluation over 2^19 values (lib.rs:173-218, the
wide-instance kernel), every Guard scalar and base must equal the CPU oracle's bit for bit, and both must reject (the pairing fails:
SingleStrategy -> ConstraintSystemFailure)."""
import os
import random

import pytest

import circuits
import oracle_lib

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CACHE = os.path.join(ROOT, "tests", "golden", "_cache")
P_MOD = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
K, N_PUB = 21, 1 << 19


class _Keys:
    """what circuits.oracle_* need of a Setup: the oracle library, VK / params bytes, the verifier options"""
    multiopen, transcript, circuit_instances, ninst_cols = 0, 0, 1, 1

    def __init__(self, vk, params):
        self.L, self.vk, self.params = oracle_lib.load(), vk, params

    def use(self):
        self.L.h2o_set_verify_options(0, 0)
        self.L.h2o_set_circuit_instances(1)


def _keys():
    os.makedirs(CACHE, exist_ok=True)
    pv, pp = os.path.join(CACHE, f"vm_k{K}_pub{N_PUB}.vk"), os.path.join(CACHE, f"vm_k{K}_pub{N_PUB}.params")
    if os.path.exists(pv) and os.path.exists(pp):
        return _Keys(open(pv, "rb").read(), open(pp, "rb").read())
    s = circuits.setup_vector_mul(K, N_PUB)
    vk, params = s.vk, s.params
    s.free()
    for path, data in ((pv, vk), (pp, params)):
        with open(path + ".tmp", "wb") as f:
            f.write(data)
        os.replace(path + ".tmp", path)
    return _Keys(vk, params)


def _random_point(rnd):
    """32 bytes that G1Affine::from_bytes accepts: x with x^3 + 3 a square, a random sign bit (byte 31 bit 6)"""
    while True:
        x = rnd.randrange(P_MOD)
        if pow((x * x * x + 3) % P_MOD, (P_MOD - 1) // 2, P_MOD) == 1:
            b = bytearray(x.to_bytes(32, "little"))
            if rnd.random() < 0.5:
                b[31] |= 0x40
            return bytes(b)


def test_two_to_the_19_public_inputs_at_k21():
    import halo2_verifier_amd as h2v
    s = _keys()
    ctx = h2v.Context(h2v.ParamsKZG(s.params, h2v.SerdeFormat.RawBytes), h2v.VerifyingKey(s.vk, h2v.SerdeFormat.RawBytes))
    shape = ctx.proof_shape()
    assert shape["proof_len"] == 1024 and shape["n_points"] == 12 and shape["n_scalars"] == 20
    rnd = random.Random(21)
    # the proof's layout by byte offset is the plan's; for this VK: 9 points, 20 scalars, the 2 multi-open points... take it from the
    # oracle instead of assuming: a byte string is a decodable proof iff the oracle's transcript reads it through to the end
    pieces = [_random_point(rnd) for _ in range(32)]
    scalars = [rnd.randrange(circuits.R_MOD).to_bytes(32, "little") for _ in range(32)]
    inst = [[rnd.randrange(circuits.R_MOD).to_bytes(32, "little") for _ in range(N_PUB)]]
    proof = None
    for n_main in range(0, 13):     # points, then scalars, then the remaining points (h1, h2): find the split the VK's transcript accepts
        cand = b"".join(pieces[:n_main]) + b"".join(scalars[:20]) + b"".join(pieces[n_main:12])
        rc, g = circuits.oracle_guard(s, cand, inst)
        if rc == 0:
            proof, expect = cand, g
            break
    assert proof is not None
    rc, got = ctx.guard_msm(proof, inst)
    assert rc == 0
    assert got["challenges"] == expect["challenges"]
    assert got["right_scalars"] == expect["right_scalars"] and got["right_bases"] == expect["right_bases"]
    assert got["left_scalars"] == expect["left_scalars"] and got["left_bases"] == expect["left_bases"]
    assert ctx.verify_each([proof], [inst]) == [-2] and circuits.oracle_verify_single(s, proof, inst) == -2
    r = [rnd.randrange(1, circuits.R_MOD)]
    res = ctx.verify_batch([proof], [inst], r)
    assert res == circuits.oracle_verify_batch(s, [proof], [inst], r) and res[0] is False and res[1] == [0]
    # one value changed at the far end of the column changes the instance evaluation, hence the Guard
    inst2 = [inst[0][:-1] + [(int.from_bytes(inst[0][-1], "little") ^ 1).to_bytes(32, "little")]]
    rc2, got2 = ctx.guard_msm(proof, inst2)
    rc3, exp2 = circuits.oracle_guard(s, proof, inst2)
    assert rc2 == 0 and rc3 == 0 and got2 == exp2 and got2["right_scalars"] != got["right_scalars"]
    ctx.close()
