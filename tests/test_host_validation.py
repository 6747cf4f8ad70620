"""The ctypes wrapper must never let the C side read past a Python buffer: every length the library will index
is checked in Python first.  No GPU needed: the checks fire before any call into the library."""
import pytest

import halo2_verifier_amd as h2v
from halo2_verifier_amd import verifier


class _Lib:
    def __getattr__(self, name):
        raise AssertionError(f"the C library must not be reached ({name})")


def _ctx():
    c = object.__new__(h2v.Context)          # no device: only the marshalling code is under test
    c._lib, c._h = _Lib(), None
    return c


S = (5).to_bytes(32, "little")


def test_instances_must_match_proofs():
    with pytest.raises(ValueError, match="instance lists"):
        _ctx().verify_batch([b"\0" * 1024] * 2, [[[S]]], [1, 2])
    with pytest.raises(ValueError, match="instance lists"):
        _ctx().verify_each([b"\0" * 1024] * 2, [[[S]]])


def test_rand_must_hold_one_scalar_per_proof():
    with pytest.raises(ValueError, match="one scalar per proof"):
        _ctx().verify_batch([b"\0" * 1024] * 2, [[[S]], [[S]]], [1])


def test_scalars_given_as_bytes_must_be_32_bytes():
    with pytest.raises(ValueError, match="exactly 32 bytes"):
        _ctx().verify_batch([b"\0" * 1024], [[[b"\1" * 31]]], [1])
    with pytest.raises(ValueError, match="exactly 32 bytes"):
        _ctx().verify_batch([b"\0" * 1024], [[[S]]], [b"\1" * 8])
    with pytest.raises(ValueError):
        verifier._scalar32(1 << 256)
    with pytest.raises(ValueError):
        verifier._scalar32(-1)
    assert verifier._scalar32(7) == (7).to_bytes(32, "little") and verifier._scalar32(bytearray(S)) == S


def test_column_count_must_agree_across_proofs():
    with pytest.raises(ValueError, match="number of instance columns"):
        _ctx().verify_batch([b"\0" * 1024] * 2, [[[S]], [[S], [S]]], [1, 2])


def test_proofs_must_be_bytes():
    with pytest.raises(TypeError):
        _ctx().verify_each(["not bytes"], [[[S]]])


def test_msm_and_pairing_argument_lengths():
    with pytest.raises(ValueError):
        _ctx().msm_g1([1, 2], [b"\0" * 64])
    with pytest.raises(ValueError):
        _ctx().msm_g1([1], [b"\0" * 63])
    with pytest.raises(ValueError):
        _ctx().pairing_check(b"\0" * 64, b"\0" * 10)


def test_batch_upload_buffer_lengths():
    b = object.__new__(h2v.Batch)
    b._lib, b._h = _Lib(), None
    with pytest.raises(ValueError, match="whole number of proofs"):
        b.upload(b"\0" * 1500, 1024, b"", [0])
    with pytest.raises(ValueError, match="instances_flat"):
        b.upload(b"\0" * 2048, 1024, b"\0" * 32, [1])
    with pytest.raises(ValueError, match="rand_tail"):
        b.upload(b"\0" * 1024, 1024, b"\0" * 32, [1], b"\0" * 33)


def test_accumulator_strategy_checks_rand_length():
    s = h2v.AccumulatorStrategy(h2v.ParamsKZG(b"\0" * 164), rand=[1])
    vk = h2v.VerifyingKey(b"vk", h2v.SerdeFormat.RawBytes)
    h2v.verify_proof(s.params, vk, s, [[S]], b"p1")
    h2v.verify_proof(s.params, vk, s, [[S]], b"p2")
    with pytest.raises(ValueError, match="one scalar per accumulated proof"):
        s.finalize()
