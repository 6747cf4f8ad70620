"""Differential fuzzing of the whole path against the CPU oracle: random single-byte corruptions anywhere in a proof
(compressed points incl. their flag bits, evaluations, the multi-open tail) and random public inputs.  Whatever the
corruption does — undecodable point, identity, non-canonical scalar, or a well-formed but wrong proof — the per-proof
plonk::Error, the accumulators and the verdicts must equal the oracle's, under both strategies and all four
multi-open x transcript instantiations.  24 seeds on the headline circuit, 6 more on the lookup / shuffle / two-phase circuits and on
two-instance transcripts."""
import random

import pytest

import circuits
from circuits import R_MOD

pytestmark = pytest.mark.gpu


def _ctx(s):
    import halo2_verifier_amd as h2v
    return h2v.Context(h2v.ParamsKZG(s.params, h2v.SerdeFormat.RawBytes), h2v.VerifyingKey(s.vk, h2v.SerdeFormat.RawBytes),
                       multiopen=s.multiopen, transcript=s.transcript)


def _mutate(rnd, P, I, share=0.75):
    P2, I2 = list(P), list(I)
    for i in range(len(P)):
        if rnd.random() > share:
            continue
        kind = rnd.random()
        if kind < 0.8:                                   # one byte of the proof
            b = bytearray(P2[i])
            pos = rnd.randrange(len(b))
            b[pos] = rnd.choice([b[pos] ^ (1 << rnd.randrange(8)), rnd.randrange(256), 0x00, 0xff, 0x80, 0x40])
            P2[i] = bytes(b)
        else:                                            # one public input
            col = list(I2[i][0])
            col[rnd.randrange(len(col))] = circuits.le32(rnd.randrange(R_MOD))
            I2[i] = [col]
    return P2, I2


INSTANTIATIONS = [(circuits.SHPLONK, circuits.BLAKE2B), (circuits.GWC, circuits.BLAKE2B), (circuits.SHPLONK, circuits.KECCAK256), (circuits.GWC, circuits.KECCAK256)]


@pytest.mark.parametrize("seed", range(1, 25))     # 24 seeds: every multi-open x transcript instantiation six times
def test_random_corruptions_match_the_oracle(seed):
    mo, trk = INSTANTIATIONS[seed % 4]
    rnd = random.Random(1000 + seed)
    s = circuits.setup_vector_mul(8, 6).set_options(mo, trk)
    P, I = circuits.prove_vector_mul_batch(s, 64, seed=seed, threads=16)
    ctx = _ctx(s)
    P2, I2 = _mutate(rnd, P, I)
    rand = [rnd.randrange(1, R_MOD) for _ in range(len(P))]
    got = ctx.verify_batch(P2, I2, rand)
    exp = circuits.oracle_verify_batch(s, P2, I2, rand)
    assert got == exp
    assert len(set(exp[1])) >= 2                         # the corruptions really produced errors of more than one kind
    each = ctx.verify_each(P2, I2)
    assert each == [circuits.oracle_verify_single(s, p, i) for p, i in zip(P2, I2)]
    assert 0 in each and -2 in each                      # accepted ones and well-formed-but-wrong ones
    ctx.close(); s.free()


@pytest.mark.parametrize("family,seed", [("wide", 31), ("wide", 32), ("shuffle", 33), ("shuffle", 34), ("wide_m2", 35), ("shuffle_m2", 36)])
def test_random_corruptions_other_circuits(family, seed):
    """The same differential fuzzing on the circuits with lookups / shuffles / rotations, the two-phase circuit with user challenges,
    and on proofs that carry two circuit instances per transcript."""
    rnd = random.Random(2000 + seed)
    mo, trk = INSTANTIATIONS[seed % 4]
    m = 2 if family.endswith("_m2") else 1
    if family.startswith("wide"):
        s = circuits.setup_wide(8, A=8, F=5, L_=1, Sh=1, deg=4)
        pairs = [circuits.prove_wide_multi(s, m, witness_seed=seed * 10 + i, rng_seed=i) for i in range(10)]
    else:
        s = circuits.setup_shuffle(8, 4, 32)
        pairs = [circuits.prove_shuffle_multi(s, m, data_seed=seed * 10 + i, rng_seed=i) for i in range(10)]
    s.set_options(mo, trk).set_circuit_instances(m)
    if mo != circuits.SHPLONK or trk != circuits.BLAKE2B:   # the proofs above were made before the options were set
        if family.startswith("wide"):
            pairs = [circuits.prove_wide_multi(s, m, witness_seed=seed * 10 + i, rng_seed=i) for i in range(10)]
        else:
            pairs = [circuits.prove_shuffle_multi(s, m, data_seed=seed * 10 + i, rng_seed=i) for i in range(10)]
    P, I = [p for p, _ in pairs], [i for _, i in pairs]
    import halo2_verifier_amd as h2v
    ctx = h2v.Context(h2v.ParamsKZG(s.params, h2v.SerdeFormat.RawBytes), h2v.VerifyingKey(s.vk, h2v.SerdeFormat.RawBytes),
                      multiopen=s.multiopen, transcript=s.transcript, circuit_instances=m)
    P2 = list(P)
    for i in range(len(P)):
        if rnd.random() < 0.7:
            b = bytearray(P2[i]); pos = rnd.randrange(len(b))
            b[pos] = rnd.choice([b[pos] ^ (1 << rnd.randrange(8)), rnd.randrange(256), 0x00, 0xff, 0x80, 0x40])
            P2[i] = bytes(b)
    rand = [rnd.randrange(1, R_MOD) for _ in P]
    assert ctx.verify_batch(P2, I, rand) == circuits.oracle_verify_batch(s, P2, I, rand)
    assert ctx.verify_each(P2, I) == [circuits.oracle_verify_single(s, p, i) for p, i in zip(P2, I)]
    assert ctx.verify_each(P, I) == [0] * len(P)
    ctx.close(); s.free()
