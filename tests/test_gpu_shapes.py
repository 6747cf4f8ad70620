"""Per-proof instance shapes and mixed VerifyingKeys under one AccumulatorStrategy (VERDICT r1 "missing" 3).

The reference's verify_proof takes `instances` and `vk` per call (lib.rs:33-49) and AccumulatorStrategy only ever sees MSMs
(kzg/strategy.rs:125-140), so one accumulation may mix instance column lengths and even VKs over the same params.  Expected
values are restated from the oracle's per-proof Guards (circuits.oracle_accumulate): proof i scaled by the product of the draws
of all later proofs in CALL order, one pairing at the end."""
import random

import pytest

import circuits
from circuits import R_MOD

pytestmark = pytest.mark.gpu


def _ctx(s):
    import halo2_verifier_amd as h2v
    return h2v.Context(h2v.ParamsKZG(s.params, h2v.SerdeFormat.RawBytes), h2v.VerifyingKey(s.vk, h2v.SerdeFormat.RawBytes))


def _mixed(s, lens, seed):
    rnd = random.Random(seed)
    P, I = [], []
    for j, m in enumerate(lens):
        a = [rnd.randrange(R_MOD) for _ in range(s.n_mul)]
        b = [rnd.randrange(R_MOD) for _ in range(m)] + [0] * (s.n_mul - m)
        p, inst = circuits.prove_vector_mul_len(s, a, b, m, rng_seed=seed * 100 + j)
        P.append(p); I.append(inst)
    return P, I


def test_batch_with_per_proof_instance_shapes():
    s = circuits.setup_vector_mul(8, 8)
    lens = [8, 5, 8, 3, 5, 8, 0, 3, 3, 8, 5]
    P, I = _mixed(s, lens, 7)
    rnd = random.Random(70)
    rand = [rnd.randrange(1, R_MOD) for _ in P]
    ctx = _ctx(s)
    got = ctx.verify_batch(P, I, rand)                      # dispatches to h2v_verify_batch_shapes
    exp = circuits.oracle_accumulate([(s, p, i) for p, i in zip(P, I)], rand)
    assert got == exp and got[0] is True
    # the same proofs one by one under SingleStrategy
    assert ctx.verify_each(P, I) == [0] * len(P)
    # a wrong public input in a minority shape: rejected, accumulators still bit-exact, SingleStrategy names the proof
    I2 = [list(map(list, i)) for i in I]
    I2[3] = [[circuits.le32(9)] + I[3][0][1:]]
    got = ctx.verify_batch(P, I2, rand)
    exp = circuits.oracle_accumulate([(s, p, i) for p, i in zip(P, I2)], rand)
    assert got == exp and got[0] is False and got[1] == [0] * len(P)
    assert ctx.verify_each(P, I2) == [0, 0, 0, -2] + [0] * (len(P) - 4)
    # a proof presented with another proof's shape (its own values truncated): rejected by the pairing only
    I3 = list(I); I3[0] = [I[0][0][:5]]
    got = ctx.verify_batch(P, I3, rand)
    assert got == circuits.oracle_accumulate([(s, p, i) for p, i in zip(P, I3)], rand) and got[0] is False
    # an undecodable point in one shape group: Transcript for that proof, nothing contributed, verdict False
    bad = list(P); b = bytearray(bad[4]); b[0:32] = b"\xff" * 32; bad[4] = bytes(b)
    got = ctx.verify_batch(bad, I, rand)
    exp = circuits.oracle_accumulate([(s, p, i) for p, i in zip(bad, I)], rand)
    assert got == exp and got[0] is False and got[1][4] == -5
    # OS-drawn multipliers
    assert ctx.verify_batch(P, I, None)[0] is True
    ctx.close()
    s.free()


def test_one_strategy_accumulates_proofs_of_different_vks():
    import halo2_verifier_amd as h2v
    s8 = circuits.setup_vector_mul(8, 8)
    s4 = circuits.setup_vector_mul(8, 4)
    assert s8.params == s4.params and s8.vk != s4.vk
    P8, I8 = circuits.prove_vector_mul_batch(s8, 3, seed=11, threads=2)
    P4, I4 = circuits.prove_vector_mul_batch(s4, 3, seed=12, threads=2)
    order = [(s8, P8[0], I8[0]), (s4, P4[0], I4[0]), (s4, P4[1], I4[1]), (s8, P8[1], I8[1]), (s8, P8[2], I8[2]), (s4, P4[2], I4[2])]
    rnd = random.Random(5)
    rand = [rnd.randrange(1, R_MOD) for _ in order]
    params = h2v.ParamsKZG(s8.params, h2v.SerdeFormat.RawBytes)
    vks = {id(s8): h2v.VerifyingKey(s8.vk, h2v.SerdeFormat.RawBytes), id(s4): h2v.VerifyingKey(s4.vk, h2v.SerdeFormat.RawBytes)}
    strat = h2v.AccumulatorStrategy(params, rand=rand)
    for s, p, i in order:
        strat = h2v.verify_proof(params, vks[id(s)], strat, i, p)
    exp = circuits.oracle_accumulate(order, rand)
    assert exp[0] is True and strat.finalize() is True
    # tamper one public input of a proof of the second VK
    bad = list(order)
    bad[2] = (s4, P4[1], [[circuits.le32(3)] + I4[1][0][1:]])
    strat = h2v.AccumulatorStrategy(params, rand=rand)
    for s, p, i in bad:
        strat = h2v.verify_proof(params, vks[id(s)], strat, i, p)
    assert circuits.oracle_accumulate(bad, rand)[0] is False and strat.finalize() is False
    s8.free(); s4.free()


def test_many_instance_shapes_go_through_a_bounded_plan_cache():
    """Instance shapes are chosen by whoever supplies the proofs, and every distinct shape compiles a plan: the
    context keeps at most 32 plans (least recently used out; a plan held by a batch stays), one call takes at most 64 distinct shapes.
    40 shapes in one accumulation — more than the cache holds — still give the oracle's result, a second pass over the same shapes
    (every plan recompiled or re-fetched) gives it again, and 65 shapes in one call are refused with H2V_ERR_UNSUPPORTED."""
    import halo2_verifier_amd as h2v
    s = circuits.setup_vector_mul(8, 70)
    lens = list(range(1, 41))
    P, I = _mixed(s, lens, 21)
    rnd = random.Random(210)
    rand = [rnd.randrange(1, R_MOD) for _ in P]
    ctx = _ctx(s)
    exp = circuits.oracle_accumulate([(s, p, i) for p, i in zip(P, I)], rand)
    assert exp[0] is True
    assert ctx.verify_batch(P, I, rand) == exp
    assert ctx.verify_batch(list(reversed(P)), list(reversed(I)), list(reversed(rand)))[0] is True
    assert ctx.verify_batch(P, I, rand) == exp
    # a staged batch holds its plan while 40 other shapes pass through the cache
    b = h2v.Batch(ctx, 1, 70)
    flat = b"".join(I[6][0])
    b.upload(P[6], len(P[6]), flat, [len(I[6][0])], rand[6].to_bytes(32, "little"))
    assert ctx.verify_batch(P, I, rand) == exp
    b.launch(with_pairing=True)
    assert b.finish()[0] is True
    b.close()
    P65, I65 = _mixed(s, list(range(0, 65)), 22)
    with pytest.raises(h2v.H2VError) as e:
        ctx.verify_batch(P65, I65, [1] * 65)
    assert e.value.code == -19
    ctx.close()
    s.free()
