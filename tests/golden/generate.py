#!/usr/bin/env python3
"""Generates tests/golden/*.json — run in the build container only (python tests/golden/generate.py).

The reference ships no proof, challenge, MSM or accept/reject vectors, and it cannot be built
here (Rust + git dependencies, no toolchain), so these vectors cannot come from the reference itself.  Inputs
(VK, params, proofs) come from the oracle's test-only prover (oracle/prover.cpp, C++); expected values come from
the independent Python restatement oracle/pyref.py.  The one piece of reference data involved is
tests/golden/kzg_bn254_8.srs (a verbatim copy of halo2_verifier/params/kzg_bn254_8.srs), used as the SRS of the
"vector_mul_reference_srs" case.  Parity of everything above the SRS level is therefore "pinned by two independent
restatements", not by reference outputs.

Compressed-point flag layout recorded in every fixture: byte 31 bit 7 = identity, bit 6 = sign(y).
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import circuits  # noqa: E402
import pyref  # noqa: E402

R = pyref.R


def case(name, s, proofs, rand_seed, notes):
    params, vk = pyref.read_params_raw(s.params), pyref.read_vk_raw(s.vk)
    mo, trk = s.multiopen, s.transcript
    M = getattr(s, "circuit_instances", 1)   # instances.len() of verify_proof (lib.rs:43): circuit instances per transcript
    out = dict(name=name, notes=notes, g1_flag_layout="byte31: bit7 identity, bit6 sign(y)", serde_format="RawBytes",
               multiopen=mo, transcript=trk, params=s.params.hex(), vk=s.vk.hex(), proofs=[])
    if M != 1:
        out["circuit_instances"] = M
    guards = []
    for label, proof, inst in proofs:
        insts = [[int.from_bytes(v, "little") for v in col] for col in inst]
        entry = dict(label=label, proof=proof.hex(), instances=[[v.hex() for v in col] for col in inst])
        try:
            g = pyref.guard(params, vk, insts, proof, mo, trk, M)
            entry["guard_status"] = 0
            entry["challenges"] = [c.to_bytes(32, "little").hex() for c in g["challenges"]]
            # the Guard term by term, in the order the reference appends them (shplonk.rs:256-264, gwc.rs:86-132)
            entry["right_scalars"] = [sc.to_bytes(32, "little").hex() for sc, _ in g["right"]]
            entry["right_bases"] = [pyref.g1_xy(b).hex() for _, b in g["right"]]
            entry["left_scalars"] = [sc.to_bytes(32, "little").hex() for sc, _ in g["left"]]
            entry["left_bases"] = [pyref.g1_xy(b).hex() for _, b in g["left"]]
            guards.append(g)
        except ValueError as e:
            entry["guard_status"] = {"transcript": -5, "opening": -4}[e.args[1]]
            guards.append(None)
        entry["single_status"] = pyref.verify_single(params, vk, insts, proof, mo, trk, M)
        out["proofs"].append(entry)
    # AccumulatorStrategy over all proofs with seeded draws: acc = sum_i (prod_{j>i} r_j) msm_i  (kzg/strategy.rs:125-136)
    rnd = random.Random(rand_seed)
    rand = [rnd.randrange(1, R) for _ in proofs]
    left, right = None, None
    for i, g in enumerate(guards):
        left, right = pyref.g1_mul(rand[i], left) if left else None, pyref.g1_mul(rand[i], right) if right else None
        if g is None:
            continue
        left = pyref.g1_add(left, pyref.msm(g["left"]))
        right = pyref.g1_add(right, pyref.msm(g["right"]))
    out["batch"] = dict(rand=[r.to_bytes(32, "little").hex() for r in rand], left=pyref.g1_xy(left).hex(), right=pyref.g1_xy(right).hex(),
                        statuses=[e["guard_status"] for e in out["proofs"]],
                        ok=bool(all(g is not None for g in guards) and pyref.pairing_check(left, right, params["s_g2"], params["g2"])))
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(out, f, indent=1)
    print(name, [e["single_status"] for e in out["proofs"]], out["batch"]["ok"])


ONLY = set(sys.argv[1:])   # python tests/golden/generate.py [case name ...] regenerates only those files


def main():
    global case
    if ONLY:
        full_case = case

        def case(name, *a, **kw):  # noqa: F811
            if name in ONLY:
                full_case(name, *a, **kw)
    if not ONLY or "wide_k16_lookup_heavy" in ONLY:
        # 6. BASELINE.json config 4: lookup-heavy VK with many advice / fixed columns at k = 16 (32 advice, 16 fixed, 8 two-column
        #    lookups, degree-5 gates; 7.8 KB proofs, 124-term Guard).  Pins omega(k=16) (poly/domain.rs:52-72) and the whole
        #    expression path at the configuration as stated.  ~70 s per proof for the test prover (2^19-point coset FFTs).
        s = circuits.setup_wide(16, A=32, F=16, L_=8, Sh=0, deg=5)
        good, inst = circuits.prove_wide(s, witness_seed=16)
        bad, inst_b = circuits.prove_wide(s, witness_seed=16, tamper=True)
        case("wide_k16_lookup_heavy", s, [("valid", good, inst), ("lookup_input_not_in_table", bad, inst_b)], 8,
             "k=16, 32 advice, 16 fixed, 8 lookups (2-column), degree 5; second proof violates a lookup")
        s.free()
        if ONLY == {"wide_k16_lookup_heavy"}:
            return
    # 7. several circuit instances per transcript (`instances.len() > 1`, lib.rs:43-55: interleaved reads :91-161, :220-253, one
    #    expression / query block per instance :273-391) — no caller inside the reference passes more than one, so these fixtures
    #    (two independent restatements agreeing) are all that pins M > 1
    s = circuits.setup_vector_mul(8, 6).set_circuit_instances(2)
    good, inst = circuits.prove_vector_mul_multi(s, 2, seed=41)
    bad_inst = [list(c) for c in inst]; bad_inst[1][0] = circuits.le32(5)             # a public input of the SECOND instance
    case("vector_mul_m2", s, [("valid", good, inst), ("second_instance_public_input", good, bad_inst)], 9,
         "two circuit instances in one transcript (SHPLONK / Blake2b), 6 multiplications each; second entry: a wrong public input of instance 1")
    s.free()
    s = circuits.setup_wide(8, A=8, F=5, L_=1, Sh=1, deg=4).set_options(circuits.GWC, circuits.BLAKE2B).set_circuit_instances(2)
    good, inst = circuits.prove_wide_multi(s, 2, witness_seed=12)
    bad, inst_b = circuits.prove_wide_multi(s, 2, witness_seed=12, tamper_at=1)
    case("wide_gwc_m2", s, [("valid", good, inst), ("lookup_violated_in_second_instance", bad, inst_b)], 10,
         "two circuit instances in one transcript (GWC / Blake2b): 8 advice, 5 fixed, 1 lookup, 1 shuffle, degree 4; second proof violates the lookup of instance 1")
    s.free()
    s = circuits.setup_shuffle(8, 4, 32).set_options(circuits.SHPLONK, circuits.KECCAK256).set_circuit_instances(3)
    good, inst = circuits.prove_shuffle_multi(s, 3, data_seed=8)
    bad, _ = circuits.prove_shuffle_multi(s, 3, data_seed=8, break_at=2)
    case("two_phase_shuffle_m3", s, [("valid", good, inst), ("broken_shuffle_in_third_instance", bad, inst)], 11,
         "three circuit instances in one transcript (SHPLONK / Keccak-256), two-phase advice with user challenges; second proof: the shuffle of instance 2 is broken")
    s.free()
    if ONLY and ONLY <= {"vector_mul_m2", "wide_gwc_m2", "two_phase_shuffle_m3"}:
        return
    # 1. the reference's vector_mul test (tests/vector_mul.rs:297-333) on the reference's own SRS file
    s = circuits.setup_vector_mul(8, 10, use_reference_srs=True)
    good, inst = circuits.prove_vector_mul(s, [2] * 10, [3] * 10, rng_seed=0)
    bad_inst = [[circuits.le32(7)] + inst[0][1:]]
    case("vector_mul_reference_srs", s, [("valid", good, inst), ("public_input_plus_one", good, bad_inst)], 1,
         "k=8, 10 multiplications 2*3=6; second entry = same proof with public_inputs[0] += 1 (must be rejected)")
    s.free()
    # 2. a batch of distinct valid proofs + malformed ones (known-s SRS)
    s = circuits.setup_vector_mul(8, 10)
    P, I = circuits.prove_vector_mul_batch(s, 4, seed=77, threads=4)
    p_bad_scalar = bytearray(P[1]); p_bad_scalar[12 * 32 + 31] = 0xff
    p_h1_identity = bytearray(P[2]); p_h1_identity[1024 - 64:1024 - 32] = b"\x00" * 31 + b"\x80"
    case("vector_mul_batch", s, [("valid0", P[0], I[0]), ("noncanonical_scalar", bytes(p_bad_scalar), I[1]), ("h1_identity", bytes(p_h1_identity), I[2]), ("valid3", P[3], I[3])], 2,
         "batch with two malformed proofs: statuses -5 (Transcript) and -4 (Opening); they contribute nothing to the accumulator")
    case("vector_mul_batch_valid", s, [(f"valid{i}", P[i], I[i]) for i in range(4)], 3, "four distinct valid proofs, one pairing")
    s.free()
    # 3. the reference's two-phase "shuffle" test (tests/shuffle.rs:272-309): user challenges, second-phase advice, no permutation argument
    s = circuits.setup_shuffle(8, 4, 32)
    good, inst = circuits.prove_shuffle(s, data_seed=5)
    bad, _ = circuits.prove_shuffle(s, data_seed=5, break_it=True)
    case("two_phase_shuffle", s, [("valid", good, inst), ("broken_shuffle", bad, inst)], 4, "W=4, H=32; second proof has two shuffled cells swapped (must be rejected)")
    s.free()
    # 4. lookup + shuffle arguments, rotations -1/0/+1, degree-5 gate
    s = circuits.setup_wide(8, A=12, F=6, L_=2, Sh=1, deg=5)
    good, inst = circuits.prove_wide(s, witness_seed=3)
    bad, inst_b = circuits.prove_wide(s, witness_seed=3, tamper=True)
    case("wide_lookup_shuffle", s, [("valid", good, inst), ("lookup_input_not_in_table", bad, inst_b)], 5,
         "12 advice, 6 fixed, 2 lookups (2-column), 1 shuffle argument, degree 5; second proof violates a lookup")
    s.free()
    # 5. the other instantiations of verify_proof's generic parameters: GWC multi-open, Keccak-256 transcript
    for mo, trk, tag in ((circuits.GWC, circuits.BLAKE2B, "gwc_blake2b"), (circuits.SHPLONK, circuits.KECCAK256, "shplonk_keccak"), (circuits.GWC, circuits.KECCAK256, "gwc_keccak")):
        s = circuits.setup_wide(8, A=8, F=5, L_=1, Sh=1, deg=4).set_options(mo, trk)
        good, inst = circuits.prove_wide(s, witness_seed=9)
        bad, inst_b = circuits.prove_wide(s, witness_seed=9, tamper=True)
        case("wide_" + tag, s, [("valid", good, inst), ("lookup_input_not_in_table", bad, inst_b)], 6, "8 advice, 5 fixed, 1 lookup, 1 shuffle, degree 4; " + tag)
        s.free()
        s = circuits.setup_shuffle(8, 4, 32).set_options(mo, trk)
        good, inst = circuits.prove_shuffle(s, data_seed=5)
        bad, _ = circuits.prove_shuffle(s, data_seed=5, break_it=True)
        case("two_phase_shuffle_" + tag, s, [("valid", good, inst), ("broken_shuffle", bad, inst)], 7, "W=4, H=32; " + tag)
        s.free()


if __name__ == "__main__":
    main()
