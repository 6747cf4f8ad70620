"""The C++ oracle against the committed golden vectors (tests/golden/*.json, produced by the independent
Python restatement oracle/pyref.py — see tests/golden/generate.py for what they do and do not pin)."""
import pytest

import circuits
import golden_util
from golden_util import h

CASES = golden_util.cases()


class _S:  # the subset of circuits.Setup the oracle helpers use
    def __init__(self, L, case):
        self.L, self.params, self.vk = L, h(case["params"]), h(case["vk"])
        self.ninst_cols = len(case["proofs"][0]["instances"])
        self.multiopen, self.transcript = case.get("multiopen", 0), case.get("transcript", 0)
        self.circuit_instances = case.get("circuit_instances", 1)

    def use(self):
        self.L.h2o_set_verify_options(self.multiopen, self.transcript)
        self.L.h2o_set_circuit_instances(self.circuit_instances)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_matches_golden(oracle, case):
    s = _S(oracle, case)
    proofs, insts = [], []
    for e in case["proofs"]:
        proof, inst = h(e["proof"]), golden_util.instances_of(e)
        proofs.append(proof); insts.append(inst)
        rc, g = circuits.oracle_guard(s, proof, inst)
        assert rc == e["guard_status"], e["label"]
        if rc == 0:
            assert [c.hex() for c in g["challenges"]] == e["challenges"]
            assert [c.hex() for c in g["right_scalars"]] == e["right_scalars"]     # term by term, reference order
            assert [c.hex() for c in g["right_bases"]] == e["right_bases"]
            assert [c.hex() for c in g["left_scalars"]] == e["left_scalars"] and [c.hex() for c in g["left_bases"]] == e["left_bases"]
        assert circuits.oracle_verify_single(s, proof, inst) == e["single_status"], e["label"]
    b = case["batch"]
    ok, st, left, right = circuits.oracle_verify_batch(s, proofs, insts, [h(r) for r in b["rand"]])
    assert (ok, st, left.hex(), right.hex()) == (b["ok"], b["statuses"], b["left"], b["right"])


def test_golden_covers_the_reference_tests_accept_reject():
    names = {c["name"]: c for c in CASES}
    vm = names["vector_mul_reference_srs"]["proofs"]     # tests/vector_mul.rs:326-330
    assert [e["single_status"] for e in vm] == [0, -2]
    sh = names["two_phase_shuffle"]["proofs"]            # tests/shuffle.rs:283-308
    assert [e["single_status"] for e in sh] == [0, -2]
