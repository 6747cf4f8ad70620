"""Host-side mirror of the reference's verifier surface over the C ABI (include/h2v.h).

Reference names kept (halo2_verifier/src/lib.rs:29-49, poly/kzg/strategy.rs:55-181,
poly/kzg/commitment.rs:22-29, plonk/vk.rs:16-26, helpers.rs:7-19, plonk/mod.rs:19-32):
``verify_proof``, ``VerifyingKey``, ``ParamsKZG``, ``SerdeFormat``, ``AccumulatorStrategy``,
``SingleStrategy``, and the ``Error`` variants as ``PlonkError``.  All arithmetic happens in the HIP
library; this file only marshals bytes.
"""
import ctypes
import enum
import os

from . import _lib
from ._lib import H2VError, check


class SerdeFormat(enum.IntEnum):  # helpers.rs:7-19
    Processed = 0
    RawBytes = 1
    RawBytesUnchecked = 2


class PlonkError(enum.IntEnum):  # plonk/mod.rs:19-32 (+ the reference's panics as one extra code)
    Ok = 0
    InvalidInstances = -1
    ConstraintSystemFailure = -2
    BoundsFailure = -3
    Opening = -4
    Transcript = -5
    InstanceTooLarge = -6
    ReferencePanic = -7


class MultiOpen(enum.IntEnum):  # the `V: Verifier` parameter of verify_proof (lib.rs:35): VerifierSHPLONK / VerifierGWC
    SHPLONK = 0
    GWC = 1


class TranscriptKind(enum.IntEnum):  # the `T: TranscriptRead` parameter (lib.rs:37): Blake2bRead / Keccak256Read
    Blake2b = 0
    Keccak256 = 1


class _Options(ctypes.Structure):
    _fields_ = [("multiopen", ctypes.c_int), ("transcript", ctypes.c_int)]


class ParamsKZG:
    """Verifier-side KZG parameters: k, g, g2, s_g2 (poly/kzg/commitment.rs:22-29) as bytes."""

    def __init__(self, data: bytes, fmt: SerdeFormat = SerdeFormat.RawBytes):
        self.data = bytes(data)
        self.format = SerdeFormat(fmt)

    @classmethod
    def read(cls, data, fmt=SerdeFormat.RawBytes):  # Params::read uses RawBytes (commitment.rs:271-278)
        return cls(data, fmt)

    @classmethod
    def from_bytes(cls, data):  # ParamsKZG::from_bytes uses Processed (commitment.rs:226-232)
        return cls(data, SerdeFormat.Processed)

    @property
    def k(self):
        return int.from_bytes(self.data[:4], "little")


class VerifyingKey:
    """VerifyingKey bytes in the reference's format (plonk/vk.rs:41-115)."""

    def __init__(self, data: bytes, fmt: SerdeFormat = SerdeFormat.RawBytes):
        self.data = bytes(data)
        self.format = SerdeFormat(fmt)

    @classmethod
    def read(cls, data, fmt):
        return cls(data, fmt)

    from_bytes = read


def _flatten_instances(instances):
    """instances: list (columns) of lists of 32-byte scalars (or ints) -> (flat bytes, col_lens)"""
    flat = bytearray()
    lens = []
    for col in instances:
        lens.append(len(col))
        for v in col:
            flat += v if isinstance(v, (bytes, bytearray)) else int(v).to_bytes(32, "little")
    return bytes(flat), lens


class Context:
    """ParamsKZG + VerifyingKey resident on one GPU (h2v_ctx)."""

    def __init__(self, params: ParamsKZG, vk: VerifyingKey = None, device: int = 0, multiopen=MultiOpen.SHPLONK,
                 transcript=TranscriptKind.Blake2b):
        self._lib = _lib.load_library()
        self._h = ctypes.c_void_p()
        vkb = vk.data if vk is not None else None
        opts = _Options(int(multiopen), int(transcript))
        check(self._lib.h2v_ctx_create_ex(params.data, len(params.data), int(params.format), vkb, len(vkb) if vkb else 0,
                                          int(vk.format) if vk is not None else 0, device, ctypes.byref(opts), ctypes.byref(self._h)))
        self.params, self.vk, self.device = params, vk, device

    def close(self):
        if self._h:
            self._lib.h2v_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- MSMKZG::eval (poly/kzg/msm.rs:81-86)
    def msm_g1(self, scalars, bases):
        """scalars: iterable of ints / 32-byte LE; bases: iterable of 64-byte x|y. -> 64-byte x|y (zeros = identity)"""
        sb = b"".join(s if isinstance(s, (bytes, bytearray)) else int(s).to_bytes(32, "little") for s in scalars)
        bb = b"".join(bases)
        n = len(sb) // 32
        if len(bb) != 64 * n:
            raise ValueError("scalars and bases differ in length")  # best_multiexp asserts equal lengths (arithmetic.rs:103)
        out = ctypes.create_string_buffer(64)
        ident = ctypes.c_int(0)
        check(self._lib.h2v_msm_g1(self._h, sb, bb, n, out, ctypes.byref(ident)))
        return out.raw

    # -- DualMSM::check (poly/kzg/msm.rs:185-203)
    def pairing_check(self, left_xy: bytes, right_xy: bytes) -> bool:
        ok = ctypes.c_int(0)
        check(self._lib.h2v_pairing_check(self._h, left_xy, right_xy, ctypes.byref(ok)))
        return bool(ok.value)

    def proof_shape(self):
        vals = [ctypes.c_size_t(0) for _ in range(5)]
        check(self._lib.h2v_ctx_proof_shape(self._h, *[ctypes.byref(v) for v in vals]))
        keys = ("proof_len", "n_points", "n_scalars", "n_right_terms", "n_instance_columns")
        return dict(zip(keys, (v.value for v in vals)))

    # -- N x verify_proof + AccumulatorStrategy::finalize
    def verify_batch(self, proofs, instances, rand=None):
        """proofs: list of bytes; instances: per proof, list of columns of scalars; rand: list of n ints/bytes or None.
        Returns (batch_ok, statuses, left_xy, right_xy)."""
        n = len(proofs)
        PA = ctypes.c_char_p * max(n, 1)
        pa = PA(*proofs) if n else PA()
        pl = (ctypes.c_size_t * max(n, 1))(*[len(p) for p in proofs])
        flats, lens = [], None
        for inst in instances:
            f, l = _flatten_instances(inst)
            if lens is not None and l != lens:
                raise ValueError("all proofs of a batch must share one instance shape")
            lens = l
            flats.append(f)
        if lens is None:  # empty batch: the column count comes from the VK
            lens = [0] * self.proof_shape()["n_instance_columns"]
        ia = PA(*flats) if n else PA()
        cl = (ctypes.c_size_t * max(len(lens), 1))(*lens)
        rb = None
        if rand is not None:
            rb = b"".join(r if isinstance(r, (bytes, bytearray)) else int(r).to_bytes(32, "little") for r in rand)
        st = (ctypes.c_int * max(n, 1))()
        ok = ctypes.c_int(0)
        left = ctypes.create_string_buffer(64)
        right = ctypes.create_string_buffer(64)
        check(self._lib.h2v_verify_batch(self._h, n, pa, pl, ia, len(lens), cl, rb, st, ctypes.byref(ok), left, right))
        return bool(ok.value), list(st)[:n], left.raw, right.raw

    def verify_each(self, proofs, instances):
        n = len(proofs)
        PA = ctypes.c_char_p * max(n, 1)
        pa = PA(*proofs) if n else PA()
        pl = (ctypes.c_size_t * max(n, 1))(*[len(p) for p in proofs])
        flats, lens = [], None
        for inst in instances:
            f, l = _flatten_instances(inst)
            lens = l
            flats.append(f)
        if lens is None:
            lens = [0] * self.proof_shape()["n_instance_columns"]
        ia = PA(*flats) if n else PA()
        cl = (ctypes.c_size_t * max(len(lens), 1))(*lens)
        st = (ctypes.c_int * max(n, 1))()
        check(self._lib.h2v_verify_each(self._h, n, pa, pl, ia, len(lens), cl, st))
        return list(st)[:n]

    def guard_msm(self, proof, instances, cap=4096):
        f, lens = _flatten_instances(instances)
        cl = (ctypes.c_size_t * max(len(lens), 1))(*lens)
        rs, rb = ctypes.create_string_buffer(32 * cap), ctypes.create_string_buffer(64 * cap)
        ls, lb = ctypes.create_string_buffer(32 * 64), ctypes.create_string_buffer(64 * 64)
        ch = ctypes.create_string_buffer(32 * 64)
        nr, nl, nc = ctypes.c_size_t(cap), ctypes.c_size_t(64), ctypes.c_size_t(64)
        rc = self._lib.h2v_guard_msm(self._h, proof, len(proof), f, len(lens), cl, rs, rb, ctypes.byref(nr), ls, lb, ctypes.byref(nl), ch, ctypes.byref(nc))
        if rc != 0:
            return rc, None
        split = lambda buf, sz, n: [buf.raw[sz * i:sz * (i + 1)] for i in range(n)]
        return 0, dict(right_scalars=split(rs, 32, nr.value), right_bases=split(rb, 64, nr.value), left_scalars=split(ls, 32, nl.value),
                       left_bases=split(lb, 64, nl.value), challenges=split(ch, 32, nc.value))


class _Strategy:
    def __init__(self, params: ParamsKZG):
        self.params = params
        self._items = []  # (vk, instances, proof)


class AccumulatorStrategy(_Strategy):
    """poly/kzg/strategy.rs:55-79,125-140: collects proofs; finalize() = one pairing for all of them."""

    def __init__(self, params, rand=None, device=0):
        super().__init__(params)
        self.rand, self.device = rand, device

    def finalize(self) -> bool:
        if not self._items:
            return True  # empty DualMSM: both channels are the identity, e(0,..)e(0,..) == 1
        vk = self._items[0][0]
        ctx = Context(self.params, vk, self.device)
        try:
            ok, statuses, _, _ = ctx.verify_batch([p for _, _, p in self._items], [i for _, i, _ in self._items], self.rand)
            return ok
        finally:
            ctx.close()


class SingleStrategy(_Strategy):
    """poly/kzg/strategy.rs:83-102,164-176: one pairing per proof, checked inside verify_proof."""

    def __init__(self, params, device=0):
        super().__init__(params)
        self.device = device


def verify_proof(params: ParamsKZG, vk: VerifyingKey, strategy, instances, proof: bytes):
    """lib.rs:33-49.  `instances` = one circuit instance: list of columns.  With SingleStrategy returns None or
    raises H2VError(code = PlonkError); with AccumulatorStrategy returns the strategy (Output = Self)."""
    if isinstance(strategy, SingleStrategy):
        ctx = Context(params, vk, strategy.device)
        try:
            st = ctx.verify_each([proof], [instances])[0]
        finally:
            ctx.close()
        if st != 0:
            raise H2VError(st, PlonkError(st).name)
        return None
    strategy._items.append((vk, instances, proof))
    return strategy


def verify_batch(params, vk, proofs, instances, rand=None, device=0):
    ctx = Context(params, vk, device)
    try:
        return ctx.verify_batch(proofs, instances, rand)
    finally:
        ctx.close()


class Batch:
    """Staged, device-resident batch (h2v_batch): upload once, launch asynchronously on its stream, finish later.
    Several batches may be in flight on one Context (one HIP stream each)."""

    STAGES = ("decompress", "transcript", "fr_program", "fold", "msm", "pairing")

    def __init__(self, ctx: Context, max_proofs: int, max_instance_values: int = 0, stream=None, groups: int = 1):
        self.ctx, self._lib = ctx, ctx._lib
        self._h = ctypes.c_void_p()
        check(self._lib.h2v_batch_create(ctx._h, max_proofs, max_instance_values, ctypes.byref(self._h)))
        self.max_proofs = max_proofs
        self.n = 0
        self.groups = 1
        if stream is not None:
            self.set_stream(stream)
        if groups != 1:
            self.set_groups(groups)

    def set_groups(self, groups: int):
        """`groups` independent AccumulatorStrategy batches per upload/launch (h2v_batch_set_groups): group g owns proofs
        [g*n/groups, (g+1)*n/groups) and the same slice of the draws; each has its own accumulators and pairing."""
        check(self._lib.h2v_batch_set_groups(self._h, groups))
        self.groups = groups

    def close(self):
        if self._h:
            self._lib.h2v_batch_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream: int):
        check(self._lib.h2v_batch_set_stream(self._h, ctypes.c_void_p(hip_stream)))

    @property
    def stream(self) -> int:
        return self._lib.h2v_batch_stream(self._h) or 0

    def upload(self, proofs_flat: bytes, proof_len: int, instances_flat: bytes, col_lens, rand_tail=None):
        """proofs_flat: n * proof_len bytes; instances_flat: n * sum(col_lens) * 32 bytes;
        rand_tail: bytes of the Fr::random draws of proofs [first, total) of the whole batch (>= n scalars) or None."""
        n = len(proofs_flat) // proof_len if proof_len else 0
        cl = (ctypes.c_size_t * max(len(col_lens), 1))(*col_lens)
        nt = len(rand_tail) // 32 if rand_tail is not None else 0
        check(self._lib.h2v_batch_upload(self._h, n, proofs_flat, proof_len, instances_flat, len(col_lens), cl, rand_tail, nt))
        self.n = n

    def launch(self, with_pairing=True):
        check(self._lib.h2v_batch_launch(self._h, 1 if with_pairing else 0))

    def export_accumulators(self, device_dst: int):
        check(self._lib.h2v_batch_export_accumulators(self._h, ctypes.c_void_p(device_dst)))

    def fold_check_enqueue(self, device_accumulators: int, n_parts: int):
        check(self._lib.h2v_batch_fold_check_enqueue(self._h, ctypes.c_void_p(device_accumulators), n_parts))

    def finish(self):
        """-> (batch_ok, statuses, left_xy, right_xy)"""
        st = (ctypes.c_int * max(self.n, 1))()
        ok = ctypes.c_int(0)
        left, right = ctypes.create_string_buffer(64), ctypes.create_string_buffer(64)
        check(self._lib.h2v_batch_finish(self._h, st, ctypes.byref(ok), left, right))
        return bool(ok.value), list(st)[:self.n], left.raw, right.raw

    def finish_groups(self):
        """-> (group_ok[groups], statuses, left_xy[groups], right_xy[groups])"""
        g = self.groups
        st = (ctypes.c_int * max(self.n, 1))()
        ok = (ctypes.c_int * g)()
        left, right = ctypes.create_string_buffer(64 * g), ctypes.create_string_buffer(64 * g)
        check(self._lib.h2v_batch_finish_groups(self._h, st, ok, left, right, g))
        return ([bool(v) for v in ok], list(st)[:self.n], [left.raw[64 * i:64 * i + 64] for i in range(g)],
                [right.raw[64 * i:64 * i + 64] for i in range(g)])

    def set_profiling(self, on=True):
        check(self._lib.h2v_batch_set_profiling(self._h, 1 if on else 0))

    def timings_ms(self):
        arr = (ctypes.c_float * 6)()
        k = self._lib.h2v_batch_timings(self._h, arr, 6)
        return dict(zip(self.STAGES, list(arr)[:k]))
