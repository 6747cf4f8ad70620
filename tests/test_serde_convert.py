"""VerifyingKey::write / ParamsKZG::write_custom in another SerdeFormat through the product's host-side converters
(h2v_vk_convert / h2v_params_convert, csrc/serde.hip) — no GPU needed.  Checked against the CPU oracle's converter (oracle/vk.hpp,
which follows the reference's READER on both sides) and, for the layout only the reference's WRITER produces, against the
independent Python reader of oracle/pyref.py."""
import ctypes
import os
import sys

import pytest

import circuits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

PROCESSED, RAW, RAW_UNCHECKED = 0, 1, 2


def _oracle_convert(fn, data, f, t):
    buf = ctypes.create_string_buffer(1 << 22)
    n = fn(data, len(data), f, t, buf, len(buf))
    assert n > 0
    return buf.raw[:n]


@pytest.fixture(scope="module")
def setups():
    ss = [circuits.setup_vector_mul(8, 10), circuits.setup_wide(8, A=8, F=5, L_=1, Sh=1, deg=3), circuits.setup_shuffle(8, 4, 32),
          circuits.setup_wide(8, A=12, F=6, L_=2, Sh=1, deg=5)]
    yield ss
    for s in ss:
        s.free()


def test_vk_converts_like_the_oracle_in_every_direction(setups, oracle):
    import halo2_verifier_amd as h2v
    for s in setups:
        raw = h2v.VerifyingKey(s.vk, h2v.SerdeFormat.RawBytes)
        proc_o = _oracle_convert(oracle.h2o_vk_convert, s.vk, RAW, PROCESSED)
        proc = raw.to_bytes(h2v.SerdeFormat.Processed, h2v.VerifyingKey.LAYOUT_READER)
        assert proc == proc_o and len(proc) < len(s.vk)
        assert raw.to_bytes(h2v.SerdeFormat.RawBytes, h2v.VerifyingKey.LAYOUT_READER) == s.vk                  # identity
        assert raw.to_bytes(h2v.SerdeFormat.RawBytesUnchecked, h2v.VerifyingKey.LAYOUT_READER) == s.vk         # written like RawBytes (helpers.rs:48-58)
        back = h2v.VerifyingKey(proc, h2v.SerdeFormat.Processed).to_bytes(h2v.SerdeFormat.RawBytes, h2v.VerifyingKey.LAYOUT_READER)
        assert back == s.vk == _oracle_convert(oracle.h2o_vk_convert, proc_o, PROCESSED, RAW)
        assert h2v.VerifyingKey(s.vk, h2v.SerdeFormat.RawBytesUnchecked).to_bytes(h2v.SerdeFormat.Processed, 1) == proc


def test_writer_layout_is_what_the_reference_writer_emits(setups):
    """lookup.rs:36-49 / shuffle.rs:70-84 write all first expressions, then all second ones; the readers (lookup.rs:51-68) take
    pairs.  LAYOUT_WRITER bytes read with the pair-wise reader must give, argument by argument, the sequence
    inputs + tables of the key — and be byte-identical to the reader layout everywhere else (same length, same other fields)."""
    import halo2_verifier_amd as h2v
    import pyref
    multi = 0
    for s in setups:
        key = h2v.VerifyingKey(s.vk, h2v.SerdeFormat.RawBytes)
        w = key.to_bytes(h2v.SerdeFormat.RawBytes, h2v.VerifyingKey.LAYOUT_WRITER)
        assert len(w) == len(s.vk)
        a, b = pyref.read_vk_raw(s.vk), pyref.read_vk_raw(w)
        for name in ("lookups", "shuffles"):
            assert len(a[name]) == len(b[name])
            for (first, second), (fw, sw) in zip(a[name], b[name]):
                assert [e for pair in zip(fw, sw) for e in pair] == first + second
                multi += len(first) > 1
        for k2 in a:
            if k2 not in ("lookups", "shuffles"):
                assert a[k2] == b[k2], k2
        if all(len(f) <= 1 for name in ("lookups", "shuffles") for f, _ in a[name]):
            assert w == s.vk                      # one pair per argument: writer and reader agree
    assert multi > 0                              # the fixtures do contain multi-expression arguments


def test_params_convert_and_contexts_accept_the_converted_bytes(setups, oracle):
    import halo2_verifier_amd as h2v
    s = setups[1]
    p = h2v.ParamsKZG(s.params, h2v.SerdeFormat.RawBytes)
    proc = p.to_bytes()                                            # ParamsKZG::to_bytes = Processed (commitment.rs:215-224)
    assert len(proc) == 164 and proc == _oracle_convert(oracle.h2o_params_convert, s.params, RAW, PROCESSED)
    assert h2v.ParamsKZG(proc, h2v.SerdeFormat.Processed).to_bytes(h2v.SerdeFormat.RawBytes) == s.params
    # the reference's own parameter file (kzg_bn254_8.srs = k | g[n] | g_lagrange[n] | g2 | s_g2 in RawBytes)
    srs = open(os.path.join(ROOT, "tests", "golden", "kzg_bn254_8.srs"), "rb").read()
    ref_params = srs[:4] + srs[4:68] + srs[-256:]
    rp = h2v.ParamsKZG(ref_params, h2v.SerdeFormat.RawBytes).to_bytes()
    assert rp == _oracle_convert(oracle.h2o_params_convert, ref_params, RAW, PROCESSED)
    assert h2v.ParamsKZG(rp, h2v.SerdeFormat.Processed).to_bytes(h2v.SerdeFormat.RawBytes) == ref_params


def test_malformed_input_is_an_error_not_a_conversion(setups):
    import halo2_verifier_amd as h2v
    s = setups[0]
    with pytest.raises(h2v.H2VError) as e:
        h2v.VerifyingKey(s.vk[:-7], h2v.SerdeFormat.RawBytes).to_bytes(h2v.SerdeFormat.Processed)
    assert e.value.code == -17
    bad = bytearray(s.vk); bad[8 + 31] |= 0xff                     # first fixed commitment: x >= p
    with pytest.raises(h2v.H2VError):
        h2v.VerifyingKey(bytes(bad), h2v.SerdeFormat.RawBytes).to_bytes(h2v.SerdeFormat.Processed)
    with pytest.raises(h2v.H2VError):
        h2v.ParamsKZG(s.params[:100], h2v.SerdeFormat.RawBytes).to_bytes()


@pytest.mark.gpu
def test_converted_keys_verify_the_same_proofs(setups):
    """A context created from the converted (Processed, reader layout) bytes gives the same accumulators as one created from the raw bytes."""
    import random
    import halo2_verifier_amd as h2v
    s = setups[1]
    pairs = [circuits.prove_wide(s, witness_seed=i) for i in range(3)]
    P, I = [p for p, _ in pairs], [i for _, i in pairs]
    rand = [random.Random(5).randrange(1, circuits.R_MOD) for _ in P]
    raw_p, raw_v = h2v.ParamsKZG(s.params, h2v.SerdeFormat.RawBytes), h2v.VerifyingKey(s.vk, h2v.SerdeFormat.RawBytes)
    proc_p = h2v.ParamsKZG(raw_p.to_bytes(), h2v.SerdeFormat.Processed)
    proc_v = h2v.VerifyingKey(raw_v.to_bytes(h2v.SerdeFormat.Processed, h2v.VerifyingKey.LAYOUT_READER), h2v.SerdeFormat.Processed)
    a = h2v.Context(raw_p, raw_v); b = h2v.Context(proc_p, proc_v)
    ra, rb = a.verify_batch(P, I, rand), b.verify_batch(P, I, rand)
    assert ra == rb == circuits.oracle_verify_batch(s, P, I, rand) and ra[0] is True
    a.close(); b.close()
