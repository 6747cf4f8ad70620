"""GPU path against the committed golden vectors (no oracle involved): challenges, Guard MSM, per-proof
SingleStrategy status, and the seeded AccumulatorStrategy result.  Bit-exact."""
import pytest

import golden_util
from golden_util import h

pytestmark = pytest.mark.gpu
CASES = golden_util.cases()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_gpu_matches_golden(case):
    import halo2_verifier_amd as h2v
    ctx = h2v.Context(h2v.ParamsKZG(h(case["params"]), h2v.SerdeFormat.RawBytes), h2v.VerifyingKey(h(case["vk"]), h2v.SerdeFormat.RawBytes),
                      multiopen=case.get("multiopen", 0), transcript=case.get("transcript", 0), circuit_instances=case.get("circuit_instances", 1))
    proofs, insts = [], []
    for e in case["proofs"]:
        proof, inst = h(e["proof"]), golden_util.instances_of(e)
        proofs.append(proof); insts.append(inst)
        rc, g = ctx.guard_msm(proof, inst)
        assert rc == e["guard_status"], e["label"]
        if rc == 0:
            assert [c.hex() for c in g["challenges"]] == e["challenges"]
            assert [c.hex() for c in g["right_scalars"]] == e["right_scalars"]
            assert [c.hex() for c in g["right_bases"]] == e["right_bases"]
            assert [c.hex() for c in g["left_scalars"]] == e["left_scalars"] and [c.hex() for c in g["left_bases"]] == e["left_bases"]
    assert ctx.verify_each(proofs, insts) == [e["single_status"] for e in case["proofs"]]
    b = case["batch"]
    ok, st, left, right = ctx.verify_batch(proofs, insts, [h(r) for r in b["rand"]])
    assert (ok, st, left.hex(), right.hex()) == (b["ok"], b["statuses"], b["left"], b["right"])
    ctx.close()
