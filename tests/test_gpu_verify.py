"""GPU parity of the batch verifier against the CPU oracle, through the C ABI.

Bar: bit-exact (integer path).  Checked per proof: the Fiat-Shamir challenges and the Guard MSM
(scalars and bases, reference term order, shplonk.rs:256-264); per batch: the two evaluated
accumulator channels for a seeded multiplier stream (kzg/strategy.rs:129), accept/reject, and the
per-proof error codes.  Negative cases follow the reference's own tests: wrong public input
(tests/vector_mul.rs:329-330) and a broken shuffle (tests/shuffle.rs:291-308)."""
import random

import pytest

import circuits
from circuits import R_MOD

pytestmark = pytest.mark.gpu


def _ctx(s):
    import halo2_verifier_amd as h2v
    return h2v.Context(h2v.ParamsKZG(s.params, h2v.SerdeFormat.RawBytes), h2v.VerifyingKey(s.vk, h2v.SerdeFormat.RawBytes),
                       multiopen=s.multiopen, transcript=s.transcript)


@pytest.fixture(scope="module")
def vm_ref_srs():
    s = circuits.setup_vector_mul(8, 10, use_reference_srs=True)
    yield s
    s.free()


@pytest.fixture(scope="module")
def vm():
    s = circuits.setup_vector_mul(8, 10)
    yield s
    s.free()


def _check_guard(ctx, s, proof, inst):
    rc_o, g_o = circuits.oracle_guard(s, proof, inst)
    rc_g, g_g = ctx.guard_msm(proof, inst)
    assert rc_o == rc_g
    if rc_o == 0:
        assert g_g["challenges"] == g_o["challenges"]
        assert g_g["right_bases"] == g_o["right_bases"]         # term by term in the reference's order (shplonk.rs:256-264, gwc.rs:86-132)
        assert g_g["right_scalars"] == g_o["right_scalars"]
        assert g_g["left_scalars"] == g_o["left_scalars"] and g_g["left_bases"] == g_o["left_bases"]
    return rc_o


def test_proof_shape_matches_survey(vm):
    ctx = _ctx(vm)
    assert ctx.proof_shape() == dict(proof_len=1024, n_points=12, n_scalars=20, n_right_terms=18, n_instance_columns=1)
    ctx.close()


@pytest.mark.parametrize("which", ["reference_srs", "known_s"])
def test_guard_and_single_vector_mul(vm, vm_ref_srs, which):
    s = vm_ref_srs if which == "reference_srs" else vm
    ctx = _ctx(s)
    proof, inst = circuits.prove_vector_mul(s, [2] * 10, [3] * 10)   # the reference test's witness: 2 * 3 = 6
    assert circuits.oracle_verify_single(s, proof, inst) == 0
    assert _check_guard(ctx, s, proof, inst) == 0
    assert ctx.verify_each([proof], [inst]) == [0]
    # public_inputs[0] += 1  (tests/vector_mul.rs:329-330)
    bad = [[circuits.le32(7)] + inst[0][1:]]
    assert circuits.oracle_verify_single(s, proof, bad) == -2
    assert _check_guard(ctx, s, proof, bad) == 0          # well-formed, only the pairing fails
    assert ctx.verify_each([proof], [bad]) == [-2]
    ok, st, _, _ = ctx.verify_batch([proof], [bad], [5])
    assert ok is False and st == [0]
    ctx.close()


def test_batch_matches_oracle(vm):
    ctx = _ctx(vm)
    P, I = circuits.prove_vector_mul_batch(vm, 40, seed=3, threads=8)
    rnd = random.Random(99)
    for n in (0, 1, 2, 7, 40):
        rand = [rnd.randrange(1, R_MOD) for _ in range(n)]
        exp = circuits.oracle_verify_batch(vm, P[:n], I[:n], rand)
        got = ctx.verify_batch(P[:n], I[:n], rand)
        assert got == exp, n
        assert got[0] is True
    ctx.close()


def test_batch_with_bad_proofs(vm):
    ctx = _ctx(vm)
    P, I = circuits.prove_vector_mul_batch(vm, 12, seed=4, threads=8)
    rnd = random.Random(5)
    rand = [rnd.randrange(1, R_MOD) for _ in range(12)]
    P = list(P)
    p3 = bytearray(P[3]); p3[40] ^= 1; P[3] = bytes(p3)                      # flips a bit of x in point 1: usually undecodable or a different point
    p5 = bytearray(P[5]); p5[12 * 32 + 31] = 0xff; P[5] = bytes(p5)          # non-canonical scalar (first eval)
    p7 = bytearray(P[7]); p7[1024 - 1] ^= 0x40; P[7] = bytes(p7)             # flips the sign of h2: decodes, wrong point
    p9 = bytearray(P[9]); p9[1024 - 64:1024 - 32] = b"\x00" * 31 + b"\x80"; P[9] = bytes(p9)  # h1 = identity encoding -> Opening
    exp = circuits.oracle_verify_batch(vm, P, I, rand)
    got = ctx.verify_batch(P, I, rand)
    assert got[1] == exp[1]
    assert got == exp
    assert got[0] is False and exp[1][5] == -5 and exp[1][9] == -4
    assert ctx.verify_each(P, I) == [circuits.oracle_verify_single(vm, p, i) for p, i in zip(P, I)]
    ctx.close()


def test_truncated_and_oversized_proofs(vm):
    ctx = _ctx(vm)
    proof, inst = circuits.prove_vector_mul(vm, [5] * 10, [9] * 10)
    assert ctx.verify_each([proof + b"\x01\x02"], [inst]) == [0]                # trailing bytes are never read
    assert ctx.verify_each([proof[:500]], [inst]) == [-5]                       # reader runs dry in the main part
    assert ctx.verify_each([proof[:1000]], [inst]) == [-4]                      # ... inside the multi-open part
    assert circuits.oracle_verify_single(vm, proof[:500], inst) == -5
    assert circuits.oracle_verify_single(vm, proof[:1000], inst) == -4
    ctx.close()


def test_invalid_instances(vm):
    import halo2_verifier_amd as h2v
    ctx = _ctx(vm)
    proof, inst = circuits.prove_vector_mul(vm, [5] * 10, [9] * 10)
    with pytest.raises(h2v.H2VError) as e:
        ctx.verify_batch([proof], [[]], [1])                                    # no instance columns: Error::InvalidInstances
    assert e.value.code == -1
    ctx.close()


def test_two_phase_shuffle_circuit():
    s = circuits.setup_shuffle(8, 4, 32)
    ctx = _ctx(s)
    good, inst = circuits.prove_shuffle(s, data_seed=5)
    bad, _ = circuits.prove_shuffle(s, data_seed=5, break_it=True)
    assert circuits.oracle_verify_single(s, good, inst) == 0
    assert circuits.oracle_verify_single(s, bad, inst) == -2
    assert _check_guard(ctx, s, good, inst) == 0
    assert _check_guard(ctx, s, bad, inst) == 0
    assert ctx.verify_each([good, bad], [inst, inst]) == [0, -2]
    rand = [11, 13]
    assert ctx.verify_batch([good, good], [inst, inst], rand) == circuits.oracle_verify_batch(s, [good, good], [inst, inst], rand)
    assert ctx.verify_batch([good, bad], [inst, inst], rand) == circuits.oracle_verify_batch(s, [good, bad], [inst, inst], rand)
    ctx.close(); s.free()


@pytest.mark.parametrize("shape", [dict(A=8, F=5, L_=1, Sh=1, deg=3), dict(A=12, F=6, L_=2, Sh=0, deg=5), dict(A=4, F=4, L_=0, Sh=0, deg=4)])
def test_wide_circuit_lookups_shuffles(shape):
    s = circuits.setup_wide(8, **shape)
    ctx = _ctx(s)
    good, inst = circuits.prove_wide(s, witness_seed=3)
    assert circuits.oracle_verify_single(s, good, inst) == 0
    assert _check_guard(ctx, s, good, inst) == 0
    if shape["L_"]:
        bad, inst_b = circuits.prove_wide(s, witness_seed=3, tamper=True)
        assert circuits.oracle_verify_single(s, bad, inst_b) == -2
        assert ctx.verify_each([good, bad], [inst, inst_b]) == [0, -2]
    rand = [3, 5, 7]
    got = ctx.verify_batch([good] * 3, [inst] * 3, rand)
    assert got == circuits.oracle_verify_batch(s, [good] * 3, [inst] * 3, rand)
    assert got[0] is True
    ctx.close(); s.free()


@pytest.mark.parametrize("mo,trk", [(circuits.GWC, circuits.BLAKE2B), (circuits.SHPLONK, circuits.KECCAK256), (circuits.GWC, circuits.KECCAK256)])
def test_gwc_and_keccak_variants(mo, trk):
    """The other instantiations of verify_proof's generic parameters (lib.rs:33-40): VerifierGWC, Keccak256Read."""
    rnd = random.Random(mo * 2 + trk)
    # vector_mul batch incl. malformed proofs
    s = circuits.setup_vector_mul(8, 10).set_options(mo, trk)
    ctx = _ctx(s)
    P, I = circuits.prove_vector_mul_batch(s, 9, seed=6, threads=8)
    plen = len(P[0])
    assert ctx.proof_shape()["proof_len"] == plen
    for i in range(3):
        assert _check_guard(ctx, s, P[i], I[i]) == 0
    rand = [rnd.randrange(1, R_MOD) for _ in range(9)]
    assert ctx.verify_batch(P, I, rand) == circuits.oracle_verify_batch(s, P, I, rand)
    P2 = list(P)
    b = bytearray(P2[2]); b[plen - 1] ^= 0x40; P2[2] = bytes(b)        # last opening point: sign flipped
    b = bytearray(P2[4]); b[13 * 32 + 31] = 0xff; P2[4] = bytes(b)      # non-canonical evaluation
    b = bytearray(P2[6]); b[plen - 32:plen] = b"\x00" * 31 + b"\x80"; P2[6] = bytes(b)   # identity encoding in the multi-open part
    exp = circuits.oracle_verify_batch(s, P2, I, rand)
    assert ctx.verify_batch(P2, I, rand) == exp and exp[1][4] == -5 and exp[1][6] == -4 and exp[0] is False
    assert ctx.verify_each(P2, I) == [circuits.oracle_verify_single(s, p, i) for p, i in zip(P2, I)]
    ctx.close(); s.free()
    # two-phase circuit with user challenges
    s = circuits.setup_shuffle(8, 4, 32).set_options(mo, trk)
    ctx = _ctx(s)
    good, inst = circuits.prove_shuffle(s, data_seed=8)
    bad, _ = circuits.prove_shuffle(s, data_seed=8, break_it=True)
    assert _check_guard(ctx, s, good, inst) == 0
    assert ctx.verify_each([good, bad], [inst, inst]) == [0, -2] == [circuits.oracle_verify_single(s, p, inst) for p in (good, bad)]
    ctx.close(); s.free()
    # lookups + shuffle argument
    s = circuits.setup_wide(8, A=12, F=6, L_=2, Sh=1, deg=5).set_options(mo, trk)
    ctx = _ctx(s)
    good, inst = circuits.prove_wide(s, witness_seed=4)
    bad, inst_b = circuits.prove_wide(s, witness_seed=4, tamper=True)
    assert _check_guard(ctx, s, good, inst) == 0
    assert ctx.verify_each([good, bad], [inst, inst_b]) == [0, -2]
    rand = [3, 5]
    assert ctx.verify_batch([good, good], [inst, inst], rand) == circuits.oracle_verify_batch(s, [good, good], [inst, inst], rand)
    ctx.close(); s.free()


def test_lookup_heavy_vk_shape():
    """BASELINE.json config 4 shape: 32 advice, 16 fixed, 8 two-column lookups, rotations -1/0/+1 on a quarter of
    the advice columns, degree-5 gates, 33 permutation columns in 11 sets.  k is kept small here (verifier work does not depend on k)."""
    s = circuits.setup_wide(9, A=32, F=16, L_=8, Sh=0, deg=5)
    ctx = _ctx(s)
    shape = ctx.proof_shape()
    assert shape["n_points"] == 32 + 3 * 8 + 11 + 1 + 4 + 2 and shape["n_scalars"] == 48 + 16 + 1 + 33 + 32 + 40
    P, I = [], []
    for seed in range(6):
        p, i = circuits.prove_wide(s, witness_seed=seed, rng_seed=100 + seed)
        P.append(p); I.append(i)
    assert _check_guard(ctx, s, P[0], I[0]) == 0
    rnd = random.Random(16)
    rand = [rnd.randrange(1, R_MOD) for _ in range(6)]
    got = ctx.verify_batch(P, I, rand)
    assert got == circuits.oracle_verify_batch(s, P, I, rand) and got[0] is True
    bad, inst_b = circuits.prove_wide(s, witness_seed=0, tamper=True)
    assert ctx.verify_each([P[0], bad], [I[0], inst_b]) == [0, -2]
    ctx.close(); s.free()


def test_bench_configuration_k14():
    """BASELINE.json config 2's circuit size: k = 14 (omega, n^-1 and the blinding rows follow k; the verifier's work does
    not), 8 public inputs, known-s SRS with the bench's seed.  Guard MSM, challenges, accumulators and verdicts against the
    oracle, plus a wrong public input."""
    s = circuits.setup_vector_mul(14, 8, s_seed=0x48325630)
    ctx = _ctx(s)
    P, I = circuits.prove_vector_mul_batch(s, 6, seed=0x48325630, threads=6)
    assert len(P[0]) == 1024
    for i in range(2):
        assert _check_guard(ctx, s, P[i], I[i]) == 0
    rnd = random.Random(14)
    rand = [rnd.randrange(1, R_MOD) for _ in range(6)]
    got = ctx.verify_batch(P, I, rand)
    assert got == circuits.oracle_verify_batch(s, P, I, rand) and got[0] is True
    I2 = list(I)
    I2[4] = [[circuits.le32(1)] + I[4][0][1:]]
    bad = ctx.verify_batch(P, I2, rand)
    assert bad == circuits.oracle_verify_batch(s, P, I2, rand) and bad[0] is False
    assert ctx.verify_each(P, I2) == [0, 0, 0, 0, -2, 0]
    ctx.close(); s.free()
