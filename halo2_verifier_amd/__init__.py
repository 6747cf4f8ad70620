"""MI355X-native Halo2/KZG/SHPLONK batch verifier — Python host side.

The compute path is the hand-written HIP library ``csrc/build/libh2v_amd.so`` (C ABI in
``include/h2v.h``).  This package only loads it and mirrors the reference's verifier surface
(``verify_proof`` / ``VerifyingKey`` / ``ParamsKZG`` / strategies, halo2_verifier/src/lib.rs:29-49)
on top of that ABI.  There is no CPU fallback: importing works anywhere, but every operation
raises ``H2VError`` unless the library and a HIP device are present.
"""
from ._lib import H2VError, lib_path, load_library, device_count  # noqa: F401
from .verifier import (  # noqa: F401
    SerdeFormat, ParamsKZG, VerifyingKey, Context, AccumulatorStrategy, SingleStrategy, verify_proof, verify_batch,
    PlonkError, Batch, MultiOpen, TranscriptKind,
)
from . import distributed  # noqa: F401
