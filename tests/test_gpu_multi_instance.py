"""Several circuit instances in ONE proof transcript: `instances.len() > 1` of the reference's verify_proof
(`instances: &[&[&[Fr]]]`, lib.rs:43-55; the interleaved reads lib.rs:91-161, 220-253; one query block per instance lib.rs:349-391).

No caller inside the reference uses it, but it is part of the signature.  Proofs come from the test prover's multi-instance
path (oracle/prover.cpp create_proof_multi); expected values from the oracle's verify_proof_multi.  Bit-exact: challenges, Guard
scalars and bases, statuses, accumulators, verdicts — for the three circuit families (vector_mul with public inputs; the
two-phase shuffle with user challenges and no instance columns; lookups + shuffle + rotations) and both multi-open schemes."""
import random

import pytest

import circuits
from circuits import R_MOD

pytestmark = pytest.mark.gpu


def _ctx(s):
    import halo2_verifier_amd as h2v
    return h2v.Context(h2v.ParamsKZG(s.params, h2v.SerdeFormat.RawBytes), h2v.VerifyingKey(s.vk, h2v.SerdeFormat.RawBytes),
                       multiopen=s.multiopen, transcript=s.transcript, circuit_instances=s.circuit_instances)


def _check_guard(ctx, s, proof, inst):
    rc, g = ctx.guard_msm(proof, inst)
    erc, eg = circuits.oracle_guard(s, proof, inst)
    assert rc == erc
    if rc == 0:
        for k in ("challenges", "left_scalars", "left_bases", "right_scalars", "right_bases"):
            assert g[k] == eg[k], k


def _family(name, M, mo=circuits.SHPLONK, tr=circuits.BLAKE2B):
    if name == "vector_mul":
        s = circuits.setup_vector_mul(8, 8)
        mk = lambda seed, bad=False: circuits.prove_vector_mul_multi(s, M, seed=seed, rng_seed=seed + 1)
    elif name == "shuffle":
        s = circuits.setup_shuffle(8, 4, 32)
        mk = lambda seed, bad=False: circuits.prove_shuffle_multi(s, M, data_seed=seed, break_at=M - 1 if bad else -1, rng_seed=seed + 1)
    else:
        s = circuits.setup_wide(8, A=8, F=5, L_=1, Sh=1, deg=4)
        mk = lambda seed, bad=False: circuits.prove_wide_multi(s, M, witness_seed=seed, tamper_at=0 if bad else -1, rng_seed=seed + 1)
    s.set_options(mo, tr).set_circuit_instances(M)
    return s, mk


@pytest.mark.parametrize("M", [2, 3])
@pytest.mark.parametrize("name", ["vector_mul", "shuffle", "wide"])
def test_multi_instance_proofs_match_the_oracle(name, M):
    s, mk = _family(name, M)
    ctx = _ctx(s)
    proofs = [mk(10 + i) for i in range(3)]
    P, I = [p for p, _ in proofs], [i for _, i in proofs]
    for p, i in proofs:
        assert circuits.oracle_verify_single(s, p, i) == 0
        _check_guard(ctx, s, p, i)
    assert ctx.verify_each(P, I) == [0, 0, 0]
    rnd = random.Random(M)
    rand = [rnd.randrange(1, R_MOD) for _ in P]
    got = ctx.verify_batch(P, I, rand)
    assert got == circuits.oracle_verify_batch(s, P, I, rand) and got[0] is True
    # one instance of one proof is wrong: a public input (vector_mul) or the witness itself (shuffle, lookup)
    if name == "vector_mul":
        bad_i = [list(c) for c in I[1]]; bad_i[M - 1][0] = circuits.le32(9)
        Pb, Ib = list(P), [I[0], bad_i, I[2]]
    else:
        pb, ib = mk(11, bad=True)
        Pb, Ib = [P[0], pb, P[2]], [I[0], ib, I[2]]
    _check_guard(ctx, s, Pb[1], Ib[1])
    assert ctx.verify_each(Pb, Ib) == [0, -2, 0] == [circuits.oracle_verify_single(s, p, i) for p, i in zip(Pb, Ib)]
    got = ctx.verify_batch(Pb, Ib, rand)
    assert got == circuits.oracle_verify_batch(s, Pb, Ib, rand) and got[0] is False and got[1] == [0, 0, 0]
    # a corrupted byte in the second instance's share of the proof: same status and accumulators as the oracle
    b = bytearray(P[0]); b[len(b) // 2] ^= 0x40; Pc = [bytes(b), P[1], P[2]]
    got = ctx.verify_batch(Pc, I, rand)
    assert got == circuits.oracle_verify_batch(s, Pc, I, rand) and got[0] is False
    ctx.close()
    s.free()


def test_multi_instance_gwc_keccak():
    s, mk = _family("wide", 2, circuits.GWC, circuits.KECCAK256)
    ctx = _ctx(s)
    p, i = mk(21)
    _check_guard(ctx, s, p, i)
    assert ctx.verify_each([p], [i]) == [0] and circuits.oracle_verify_single(s, p, i) == 0
    pb, ib = mk(21, bad=True)
    assert ctx.verify_each([pb], [ib]) == [-2]
    rand = [5, 7]
    got = ctx.verify_batch([p, pb], [i, ib], rand)
    assert got == circuits.oracle_verify_batch(s, [p, pb], [i, ib], rand) and got[0] is False
    ctx.close()
    s.free()


def test_multi_instance_shape_checks_and_strategy_surface():
    import halo2_verifier_amd as h2v
    s, mk = _family("vector_mul", 2)
    p, i = mk(31)
    ctx = _ctx(s)
    with pytest.raises(h2v.H2VError) as e:      # one instance's columns for a two-instance context (lib.rs:51-55)
        ctx.verify_each([p], [i[:1]])
    assert e.value.code == h2v.PlonkError.InvalidInstances
    assert ctx.proof_shape()["n_instance_columns"] == 2
    ctx.close()
    # the same proof presented as a one-instance proof: rejected (its transcript holds two instances)
    s1 = circuits.setup_vector_mul(8, 8)
    c1 = h2v.Context(h2v.ParamsKZG(s1.params, h2v.SerdeFormat.RawBytes), h2v.VerifyingKey(s1.vk, h2v.SerdeFormat.RawBytes))
    assert c1.verify_each([p], [i[:1]]) != [0]
    c1.close(); s1.free()
    params, vk = h2v.ParamsKZG(s.params, h2v.SerdeFormat.RawBytes), h2v.VerifyingKey(s.vk, h2v.SerdeFormat.RawBytes)
    assert h2v.verify_proof(params, vk, h2v.SingleStrategy(params, circuit_instances=2), i, p) is None
    st = h2v.AccumulatorStrategy(params, rand=[3, 4], circuit_instances=2)
    st = h2v.verify_proof(params, vk, st, i, p)
    p2, i2 = mk(32)
    st = h2v.verify_proof(params, vk, st, i2, p2)
    assert st.finalize() is True
    s.free()
