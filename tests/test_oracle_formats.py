"""Byte formats of the reference (helpers.rs:7-166, plonk/vk.rs:41-115, kzg/commitment.rs:142-207) through the oracle:
RawBytes <-> Processed round trips of params and VKs, and verification with Processed inputs."""
import ctypes

import circuits


def _convert(fn, data, f, t):
    buf = ctypes.create_string_buffer(1 << 22)
    n = fn(data, len(data), f, t, buf, len(buf))
    assert n > 0
    return buf.raw[:n]


def test_params_and_vk_round_trip_between_serde_formats(oracle):
    s = circuits.setup_wide(8, A=8, F=5, L_=1, Sh=1, deg=3)
    p_proc = _convert(oracle.h2o_params_convert, s.params, 1, 0)
    assert len(p_proc) == 164                                     # 4 + 32 + 64 + 64 (kzg/commitment.rs:209-213)
    assert _convert(oracle.h2o_params_convert, p_proc, 0, 1) == s.params
    v_proc = _convert(oracle.h2o_vk_convert, s.vk, 1, 0)
    assert len(v_proc) < len(s.vk)
    assert _convert(oracle.h2o_vk_convert, v_proc, 0, 1) == s.vk
    # verification is format-independent
    proof, inst = circuits.prove_wide(s, witness_seed=3)
    f = b"".join(b"".join(c) for c in inst)
    cl = (ctypes.c_size_t * 1)(8)
    oracle.h2o_set_verify_options(0, 0)   # SHPLONK + Blake2b (the options are thread-local state on the C side)
    assert oracle.h2o_verify_single(p_proc, len(p_proc), 0, v_proc, len(v_proc), 0, f, cl, 1, proof, len(proof)) == 0
    s.free()


def test_truncated_vk_is_rejected(oracle):
    s = circuits.setup_vector_mul(8, 10)
    buf = ctypes.create_string_buffer(1 << 16)
    assert oracle.h2o_vk_convert(s.vk[:-5], len(s.vk) - 5, 1, 0, buf, len(buf)) == 0
    s.free()
