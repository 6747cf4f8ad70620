"""ctypes loader for the CPU oracle (oracle/build/liboracle.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None
c = ctypes


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(ROOT, "oracle", "build", "liboracle.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    L = c.CDLL(path)
    szp = c.POINTER(c.c_size_t)
    intp = c.POINTER(c.c_int)
    L.h2o_fr_from_uniform.argtypes = [c.c_char_p, c.c_char_p]
    L.h2o_blake2b_personal.argtypes = [c.c_char_p, c.c_char_p, c.c_size_t, c.c_char_p]
    L.h2o_g1_decompress.argtypes = [c.c_char_p, c.c_char_p, intp]
    L.h2o_g1_compress.argtypes = [c.c_char_p, c.c_char_p]
    L.h2o_g1_msm.argtypes = [c.c_char_p, c.c_char_p, c.c_size_t, c.c_char_p, intp]
    L.h2o_pairing_check.argtypes = [c.c_char_p, c.c_size_t, c.c_int, c.c_char_p, c.c_char_p, intp]
    L.h2o_pairing_product_is_one.argtypes = [c.c_char_p, c.c_char_p, c.c_char_p, c.c_char_p, intp]
    L.h2o_params_convert.restype = c.c_size_t
    L.h2o_params_convert.argtypes = [c.c_char_p, c.c_size_t, c.c_int, c.c_int, c.c_char_p, c.c_size_t]
    L.h2o_vk_convert.restype = c.c_size_t
    L.h2o_vk_convert.argtypes = [c.c_char_p, c.c_size_t, c.c_int, c.c_int, c.c_char_p, c.c_size_t]
    L.h2o_verify_single.argtypes = [c.c_char_p, c.c_size_t, c.c_int, c.c_char_p, c.c_size_t, c.c_int, c.c_char_p, szp, c.c_size_t, c.c_char_p, c.c_size_t]
    L.h2o_guard_msm.argtypes = [c.c_char_p, c.c_size_t, c.c_int, c.c_char_p, c.c_size_t, c.c_int, c.c_char_p, szp, c.c_size_t, c.c_char_p, c.c_size_t,
                                c.c_char_p, c.c_char_p, szp, c.c_char_p, c.c_char_p, szp, c.c_char_p, szp]
    L.h2o_verify_batch.argtypes = [c.c_char_p, c.c_size_t, c.c_int, c.c_char_p, c.c_size_t, c.c_int, c.c_size_t, c.c_char_p, c.c_size_t, c.c_char_p, szp, c.c_size_t,
                                   c.c_char_p, intp, intp, c.c_char_p, c.c_char_p]
    L.h2o_verify_each.argtypes = [c.c_char_p, c.c_size_t, c.c_int, c.c_char_p, c.c_size_t, c.c_int, c.c_size_t, c.c_char_p, c.c_size_t, c.c_char_p, szp, c.c_size_t, intp]
    for f in ("h2o_setup_vector_mul", "h2o_setup_shuffle", "h2o_setup_wide"):
        getattr(L, f).restype = c.c_void_p
    L.h2o_setup_vector_mul.argtypes = [c.c_uint32, c.c_size_t, c.c_char_p, c.c_size_t, c.c_uint64]
    L.h2o_setup_shuffle.argtypes = [c.c_uint32, c.c_size_t, c.c_size_t, c.c_char_p, c.c_size_t, c.c_uint64]
    L.h2o_setup_wide.argtypes = [c.c_uint32, c.c_size_t, c.c_size_t, c.c_size_t, c.c_size_t, c.c_uint32, c.c_uint64, c.c_char_p, c.c_size_t, c.c_uint64]
    L.h2o_setup_free.argtypes = [c.c_void_p]
    L.h2o_setup_set_options.argtypes = [c.c_void_p, c.c_int, c.c_int]
    L.h2o_setup_set_options.restype = None
    L.h2o_set_verify_options.argtypes = [c.c_int, c.c_int]
    L.h2o_set_verify_options.restype = None
    L.h2o_keccak256.argtypes = [c.c_char_p, c.c_size_t, c.c_char_p]
    L.h2o_setup_vk.restype = c.c_size_t
    L.h2o_setup_vk.argtypes = [c.c_void_p, c.c_int, c.c_char_p, c.c_size_t]
    L.h2o_setup_params.restype = c.c_size_t
    L.h2o_setup_params.argtypes = [c.c_void_p, c.c_int, c.c_char_p, c.c_size_t]
    L.h2o_prove_vector_mul.restype = c.c_size_t
    L.h2o_prove_vector_mul.argtypes = [c.c_void_p, c.c_char_p, c.c_char_p, c.c_uint64, c.c_char_p, c.c_size_t, c.c_char_p]
    L.h2o_prove_vector_mul_len.restype = c.c_size_t
    L.h2o_prove_vector_mul_len.argtypes = [c.c_void_p, c.c_char_p, c.c_char_p, c.c_size_t, c.c_uint64, c.c_char_p, c.c_size_t, c.c_char_p]
    L.h2o_set_circuit_instances.argtypes = [c.c_size_t]
    L.h2o_set_circuit_instances.restype = None
    L.h2o_prove_vector_mul_multi.restype = c.c_size_t
    L.h2o_prove_vector_mul_multi.argtypes = [c.c_void_p, c.c_size_t, c.c_char_p, c.c_char_p, c.c_uint64, c.c_char_p, c.c_size_t, c.c_char_p]
    L.h2o_prove_shuffle_multi.restype = c.c_size_t
    L.h2o_prove_shuffle_multi.argtypes = [c.c_void_p, c.c_size_t, c.c_uint64, c.c_int, c.c_uint64, c.c_char_p, c.c_size_t]
    L.h2o_prove_wide_multi.restype = c.c_size_t
    L.h2o_prove_wide_multi.argtypes = [c.c_void_p, c.c_size_t, c.c_uint64, c.c_int, c.c_uint64, c.c_char_p, c.c_size_t, c.c_char_p]
    L.h2o_prove_vector_mul_batch.restype = c.c_size_t
    L.h2o_prove_vector_mul_batch.argtypes = [c.c_void_p, c.c_size_t, c.c_uint64, c.c_uint, c.c_char_p, c.c_size_t, c.c_char_p]
    L.h2o_prove_shuffle.restype = c.c_size_t
    L.h2o_prove_shuffle.argtypes = [c.c_void_p, c.c_uint64, c.c_int, c.c_uint64, c.c_char_p, c.c_size_t]
    L.h2o_prove_wide.restype = c.c_size_t
    L.h2o_prove_wide.argtypes = [c.c_void_p, c.c_uint64, c.c_int, c.c_uint64, c.c_char_p, c.c_size_t, c.c_char_p]
    _LIB = L
    return L


def g1_msm(L, scalars, bases):
    """scalars: list of ints; bases: list of 64-byte x|y. Returns 64-byte x|y."""
    n = len(scalars)
    sb = b"".join(int(s).to_bytes(32, "little") for s in scalars)
    bb = b"".join(bases)
    out = c.create_string_buffer(64)
    ident = c.c_int(0)
    rc = L.h2o_g1_msm(sb, bb, n, out, c.byref(ident))
    assert rc == 0, rc
    return out.raw
