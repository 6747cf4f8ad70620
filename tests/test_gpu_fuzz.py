"""Differential fuzzing of the whole path against the CPU oracle: random single-byte corruptions anywhere in a proof
(compressed points incl. their flag bits, evaluations, the multi-open tail) and random public inputs.  Whatever the
corruption does — undecodable point, identity, non-canonical scalar, or a well-formed but wrong proof — the per-proof
plonk::Error, the accumulators and the verdicts must equal the oracle's, under both strategies and all four
multi-open x transcript instantiations."""
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


@pytest.mark.parametrize("mo,trk,seed", [(circuits.SHPLONK, circuits.BLAKE2B, 1), (circuits.SHPLONK, circuits.BLAKE2B, 2),
                                         (circuits.GWC, circuits.BLAKE2B, 3), (circuits.SHPLONK, circuits.KECCAK256, 4), (circuits.GWC, circuits.KECCAK256, 5)])
def test_random_corruptions_match_the_oracle(mo, trk, seed):
    rnd = random.Random(1000 + seed)
    s = circuits.setup_vector_mul(8, 6).set_options(mo, trk)
    P, I = circuits.prove_vector_mul_batch(s, 96, seed=seed, threads=16)
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
