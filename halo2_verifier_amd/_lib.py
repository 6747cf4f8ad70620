"""ctypes binding of include/h2v.h.  Fails loudly when the HIP library is missing."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class H2VError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"h2v error {code}: {message}")
        self.code = code


def lib_path():
    return os.path.join(_HERE, "csrc", "build", "libh2v_amd.so")


c_u8p = ctypes.c_char_p
c_sz = ctypes.c_size_t
c_int = ctypes.c_int
c_vp = ctypes.c_void_p
c_szp = ctypes.POINTER(ctypes.c_size_t)
c_intp = ctypes.POINTER(ctypes.c_int)

ABI_VERSION = 3   # include/h2v.h H2V_ABI_VERSION: the struct layouts this binding mirrors

# every symbol include/h2v.h declares: (restype, argtypes)
SIGNATURES = {
    "h2v_device_count": (c_int, []),
    "h2v_last_error": (ctypes.c_char_p, []),
    "h2v_ctx_create": (c_int, [c_u8p, c_sz, c_int, c_u8p, c_sz, c_int, c_int, ctypes.POINTER(c_vp)]),
    "h2v_ctx_create_ex": (c_int, [c_u8p, c_sz, c_int, c_u8p, c_sz, c_int, c_int, ctypes.c_void_p, ctypes.POINTER(c_vp)]),
    "h2v_ctx_destroy": (None, [c_vp]),
    "h2v_abi_version": (c_int, []),
    "h2v_ctx_set_tuning": (c_int, [c_vp, ctypes.c_void_p]),
    "h2v_vk_convert": (c_int, [c_u8p, c_sz, c_int, c_int, c_int, c_u8p, c_szp]),
    "h2v_params_convert": (c_int, [c_u8p, c_sz, c_int, c_int, c_u8p, c_szp]),
    "h2v_ctx_proof_shape": (c_int, [c_vp, c_szp, c_szp, c_szp, c_szp, c_szp]),
    "h2v_msm_g1": (c_int, [c_vp, c_u8p, c_u8p, c_sz, c_u8p, c_intp]),
    "h2v_pairing_check": (c_int, [c_vp, c_u8p, c_u8p, c_intp]),
    "h2v_verify_batch": (c_int, [c_vp, c_sz, ctypes.POINTER(c_u8p), c_szp, ctypes.POINTER(c_u8p), c_sz, c_szp, c_u8p, c_intp, c_intp, c_u8p, c_u8p]),
    "h2v_verify_batch_shapes": (c_int, [c_vp, c_sz, ctypes.POINTER(c_u8p), c_szp, ctypes.POINTER(c_u8p), c_sz, c_szp, c_u8p, c_intp, c_intp, c_u8p, c_u8p]),
    "h2v_verify_batch_seeded": (c_int, [c_vp, c_sz, ctypes.POINTER(c_u8p), c_szp, ctypes.POINTER(c_u8p), c_sz, c_szp, c_u8p, c_u8p, c_u8p, c_sz, c_u8p, c_u8p, c_sz,
                                        c_intp, c_intp, c_u8p, c_u8p]),
    "h2v_verify_each": (c_int, [c_vp, c_sz, ctypes.POINTER(c_u8p), c_szp, ctypes.POINTER(c_u8p), c_sz, c_szp, c_intp]),
    "h2v_guard_msm": (c_int, [c_vp, c_u8p, c_sz, c_u8p, c_sz, c_szp, c_u8p, c_u8p, c_szp, c_u8p, c_u8p, c_szp, c_u8p, c_szp]),
    "h2v_random_scalars": (c_int, [c_u8p, c_sz]),
    "h2v_batch_create": (c_int, [c_vp, c_sz, c_sz, ctypes.POINTER(c_vp)]),
    "h2v_batch_destroy": (None, [c_vp]),
    "h2v_batch_upload": (c_int, [c_vp, c_sz, c_u8p, c_sz, c_u8p, c_sz, c_szp, c_u8p, c_sz]),
    "h2v_batch_launch": (c_int, [c_vp, c_int]),
    "h2v_batch_upload_launch": (c_int, [c_vp, c_sz, c_u8p, c_sz, c_u8p, c_sz, c_szp, c_u8p, c_sz, c_int]),
    "h2v_batch_finish": (c_int, [c_vp, c_intp, c_intp, c_u8p, c_u8p]),
    "h2v_batch_set_groups": (c_int, [c_vp, c_sz]),
    "h2v_batch_finish_groups": (c_int, [c_vp, c_intp, c_intp, c_u8p, c_u8p, c_sz]),
    "h2v_batch_accumulators": (c_int, [c_vp, ctypes.POINTER(c_vp), c_szp]),
    "h2v_batch_stream": (c_vp, [c_vp]),
    "h2v_batch_set_stream": (c_int, [c_vp, c_vp]),
    "h2v_batch_export_accumulators": (c_int, [c_vp, c_vp]),
    "h2v_batch_fold_check_enqueue": (c_int, [c_vp, c_vp, c_sz]),
    "h2v_fold_check": (c_int, [c_vp, c_vp, c_sz, c_intp, c_u8p, c_u8p]),
    "h2v_batch_timings": (c_int, [c_vp, ctypes.POINTER(ctypes.c_float), c_int]),
    "h2v_batch_set_profiling": (c_int, [c_vp, c_int]),
}


def load_library():
    """Load libh2v_amd.so (built by __graft_entry__.build() / csrc/Makefile)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64 under the same soname as the
    # system one this library links against.  Whichever is loaded first serves both, and PyTorch cannot enumerate
    # GPUs on top of the system copy ("No HIP GPUs are available").  If PyTorch is installed, let it load its runtime first.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    path = lib_path()
    if not os.path.exists(path):
        raise H2VError(-18, f"{path} not found: build it with `make -C halo2_verifier_amd/csrc` "
                            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here means the .so is stale
        fn.restype = res
        fn.argtypes = args
    if lib.h2v_abi_version() != ABI_VERSION:
        raise H2VError(-16, f"{path} has ABI version {lib.h2v_abi_version()}, this binding was written for {ABI_VERSION}: rebuild the library")
    _LIB = lib
    return lib


def last_error():
    return load_library().h2v_last_error().decode("utf-8", "replace")


def check(rc):
    if rc != 0:
        raise H2VError(rc, last_error())


def device_count():
    return load_library().h2v_device_count()
