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
from . import distributed
from ._lib import H2VError, check

_FR_MODULUS = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001


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


class _Options(ctypes.Structure):   # h2v_options
    _fields_ = [("struct_size", ctypes.c_size_t), ("multiopen", ctypes.c_int), ("transcript", ctypes.c_int), ("circuit_instances", ctypes.c_int),
                ("instance_kernel_threshold", ctypes.c_int)]


class _Tuning(ctypes.Structure):    # h2v_tuning (debug / test): forced kernel variants, 0 = automatic
    _fields_ = [("struct_size", ctypes.c_size_t)] + [(k, ctypes.c_int) for k in (
        "frvm_streams", "frvm_lds_kb", "msm_parts", "msm_global_sort", "msm_no_term_split", "msm_window_threads", "msm_window_wpw",
        "msm_window_slots", "msm_acc_waves", "pairing_one_stream", "upload_mode")]


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

    def to_bytes(self, fmt: SerdeFormat = SerdeFormat.Processed) -> bytes:
        """ParamsKZG::write_custom / to_bytes in another SerdeFormat (kzg/commitment.rs:142-152, 215-224; to_bytes = Processed)."""
        lib = _lib.load_library()
        n = ctypes.c_size_t(0)
        check(lib.h2v_params_convert(self.data, len(self.data), int(self.format), int(fmt), None, ctypes.byref(n)))
        buf = ctypes.create_string_buffer(n.value)
        check(lib.h2v_params_convert(self.data, len(self.data), int(self.format), int(fmt), buf, ctypes.byref(n)))
        return buf.raw[: n.value]

    def write(self, fmt: SerdeFormat = SerdeFormat.RawBytes) -> bytes:
        return self.to_bytes(fmt)


class VerifyingKey:
    """VerifyingKey bytes in the reference's format (plonk/vk.rs:41-115)."""

    def __init__(self, data: bytes, fmt: SerdeFormat = SerdeFormat.RawBytes):
        self.data = bytes(data)
        self.format = SerdeFormat(fmt)

    @classmethod
    def read(cls, data, fmt):
        return cls(data, fmt)

    from_bytes = read

    LAYOUT_WRITER, LAYOUT_READER = 0, 1

    def to_bytes(self, fmt: SerdeFormat, layout: int = 0) -> bytes:
        """VerifyingKey::write / to_bytes (plonk/vk.rs:41-64, 118-123) in `fmt`.  layout: LAYOUT_WRITER = exactly what the reference's
        writer emits; LAYOUT_READER = what its reader consumes — they differ for lookup / shuffle arguments of more than one
        expression pair (include/h2v.h, h2v_vk_convert)."""
        lib = _lib.load_library()
        n = ctypes.c_size_t(0)
        check(lib.h2v_vk_convert(self.data, len(self.data), int(self.format), int(fmt), int(layout), None, ctypes.byref(n)))
        buf = ctypes.create_string_buffer(n.value)
        check(lib.h2v_vk_convert(self.data, len(self.data), int(self.format), int(fmt), int(layout), buf, ctypes.byref(n)))
        return buf.raw[: n.value]

    write = to_bytes


def _scalar32(v) -> bytes:
    """One field element at the boundary: 32 little-endian bytes.  The C side reads exactly 32 bytes per scalar, so a
    shorter bytes object would make it read past the Python buffer: reject it here."""
    if isinstance(v, (bytes, bytearray, memoryview)):
        if len(v) != 32:
            raise ValueError(f"a scalar given as bytes must be exactly 32 bytes, got {len(v)}")
        return bytes(v)
    v = int(v)
    if v < 0 or v >> 256:
        raise ValueError("a scalar given as an integer must be in [0, 2^256)")
    return v.to_bytes(32, "little")


def _flatten_instances(instances):
    """instances: list (columns) of lists of 32-byte scalars (or ints) -> (flat bytes, col_lens)"""
    flat = bytearray()
    lens = []
    for col in instances:
        lens.append(len(col))
        for v in col:
            flat += _scalar32(v)
    return bytes(flat), lens


def _marshal_batch(ctx, proofs, instances):
    """Pointer arrays for h2v_verify_batch / h2v_verify_each.  Everything the C side will index is checked here: one
    instance list per proof, proofs are bytes, every scalar is 32 bytes.  Returns (n, proof ptrs, proof lens, instance
    ptrs, per-proof column lengths [n][ncols], keep-alive list)."""
    n = len(proofs)
    if len(instances) != n:
        raise ValueError(f"{n} proofs but {len(instances)} instance lists: verify_proof takes one per proof (lib.rs:33-49)")
    for p in proofs:
        if not isinstance(p, (bytes, bytearray)):
            raise TypeError("proofs must be bytes")
    PA = ctypes.c_char_p * max(n, 1)
    pa = PA(*[bytes(p) for p in proofs]) if n else PA()
    pl = (ctypes.c_size_t * max(n, 1))(*[len(p) for p in proofs])
    flats, shapes = [], []
    for inst in instances:
        f, l = _flatten_instances(inst)
        flats.append(f)
        shapes.append(l)
    ncols = len(shapes[0]) if shapes else ctx.proof_shape()["n_instance_columns"]
    for l in shapes:
        if len(l) != ncols:
            raise ValueError("all proofs of a batch must have the same number of instance columns")
    ia = PA(*flats) if n else PA()
    return n, pa, pl, ia, shapes, ncols, flats


class Context:
    """ParamsKZG + VerifyingKey resident on one GPU (h2v_ctx)."""

    def __init__(self, params: ParamsKZG, vk: VerifyingKey = None, device: int = 0, multiopen=MultiOpen.SHPLONK,
                 transcript=TranscriptKind.Blake2b, circuit_instances: int = 1, instance_kernel_threshold: int = 0):
        """circuit_instances = len(instances) of the reference's verify_proof (`instances: &[&[&[Fr]]]`, lib.rs:43): how many
        circuit instances share one proof transcript.  With M > 1 the `instances` of a proof is the list of its M x columns,
        instance by instance."""
        self._lib = _lib.load_library()
        self._h = ctypes.c_void_p()
        vkb = vk.data if vk is not None else None
        opts = _Options(ctypes.sizeof(_Options), int(multiopen), int(transcript), int(circuit_instances), int(instance_kernel_threshold))
        self.circuit_instances = int(circuit_instances)
        check(self._lib.h2v_ctx_create_ex(params.data, len(params.data), int(params.format), vkb, len(vkb) if vkb else 0,
                                          int(vk.format) if vk is not None else 0, device, ctypes.byref(opts), ctypes.byref(self._h)))
        self.params, self.vk, self.device = params, vk, device

    def close(self):
        if self._h:
            for b in list(getattr(self, "_batches", ())):   # h2v_ctx_destroy: the context's batches go first
                b.close()
            self._lib.h2v_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_tuning(self, **fields):
        """Debug / test (h2v_ctx_set_tuning): force kernel variants the library otherwise chooses from the launch shape, e.g.
        set_tuning(frvm_streams=2, msm_window_slots=3).  No arguments = back to automatic.  Not synchronised with launches in
        flight: call while the context is idle."""
        if not fields:
            check(self._lib.h2v_ctx_set_tuning(self._h, None))
            return
        t = _Tuning(ctypes.sizeof(_Tuning))
        for k, val in fields.items():
            if k == "struct_size" or not hasattr(t, k):
                raise ValueError(f"unknown tuning field {k!r}")
            setattr(t, k, int(val))
        check(self._lib.h2v_ctx_set_tuning(self._h, ctypes.byref(t)))

    # -- MSMKZG::eval (poly/kzg/msm.rs:81-86)
    def msm_g1(self, scalars, bases):
        """scalars: iterable of ints / 32-byte LE; bases: iterable of 64-byte x|y. -> 64-byte x|y (zeros = identity)"""
        sb = b"".join(_scalar32(s) for s in scalars)
        bases = list(bases)
        if any(len(b) != 64 for b in bases):
            raise ValueError("every base must be 64 bytes (x | y)")
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
        if len(left_xy) != 64 or len(right_xy) != 64:
            raise ValueError("points are 64 bytes (x | y)")
        check(self._lib.h2v_pairing_check(self._h, left_xy, right_xy, ctypes.byref(ok)))
        return bool(ok.value)

    def proof_shape(self):
        vals = [ctypes.c_size_t(0) for _ in range(5)]
        check(self._lib.h2v_ctx_proof_shape(self._h, *[ctypes.byref(v) for v in vals]))
        keys = ("proof_len", "n_points", "n_scalars", "n_right_terms", "n_instance_columns")
        return dict(zip(keys, (v.value for v in vals)))

    # -- N x verify_proof + AccumulatorStrategy::finalize
    def verify_batch(self, proofs, instances, rand=None, seed=None):
        """proofs: list of bytes; instances: per proof, list of columns of scalars (column lengths may differ from proof to
        proof, as N independent verify_proof calls allow); rand: list of n ints/bytes or None.
        seed: an existing accumulator to start from — AccumulatorStrategy::with (kzg/strategy.rs:75-78) — as
        ((left_scalars, left_bases), (right_scalars, right_bases)): scalars ints / 32-byte strings, bases 64-byte x | y.
        Returns (batch_ok, statuses, left_xy, right_xy)."""
        n, pa, pl, ia, shapes, ncols, _keep = _marshal_batch(self, proofs, instances)
        rb = None
        if rand is not None:
            if len(rand) != n:
                raise ValueError(f"rand must hold one scalar per proof ({n}), got {len(rand)}")   # the C side reads n * 32 bytes
            rb = b"".join(_scalar32(r) for r in rand)
        st = (ctypes.c_int * max(n, 1))()
        ok = ctypes.c_int(0)
        left = ctypes.create_string_buffer(64)
        right = ctypes.create_string_buffer(64)
        if seed is not None:
            if not all(l == shapes[0] for l in shapes):
                raise ValueError("a seeded batch takes one instance shape")
            sides = []
            for scalars, bases in seed:
                scalars, bases = list(scalars), list(bases)
                if len(scalars) != len(bases):
                    raise ValueError("seed scalars and bases differ in length")   # MSMKZG keeps them parallel (msm.rs:17-24)
                if any(len(b) != 64 for b in bases):
                    raise ValueError("every seed base must be 64 bytes (x | y)")
                sides.append((b"".join(_scalar32(x) for x in scalars), b"".join(bases), len(scalars)))
            lens = shapes[0] if shapes else [0] * ncols
            cl = (ctypes.c_size_t * max(ncols, 1))(*lens)
            check(self._lib.h2v_verify_batch_seeded(self._h, n, pa, pl, ia, ncols, cl, rb, sides[0][0], sides[0][1], sides[0][2], sides[1][0], sides[1][1], sides[1][2],
                                                    st, ctypes.byref(ok), left, right))
        elif all(l == shapes[0] for l in shapes):
            lens = shapes[0] if shapes else [0] * ncols
            cl = (ctypes.c_size_t * max(ncols, 1))(*lens)
            check(self._lib.h2v_verify_batch(self._h, n, pa, pl, ia, ncols, cl, rb, st, ctypes.byref(ok), left, right))
        else:
            cl = (ctypes.c_size_t * max(n * ncols, 1))(*[v for l in shapes for v in l])
            check(self._lib.h2v_verify_batch_shapes(self._h, n, pa, pl, ia, ncols, cl, rb, st, ctypes.byref(ok), left, right))
        return bool(ok.value), list(st)[:n], left.raw, right.raw

    def verify_each(self, proofs, instances):
        n, pa, pl, ia, shapes, ncols, _keep = _marshal_batch(self, proofs, instances)
        st = (ctypes.c_int * max(n, 1))()
        if all(l == shapes[0] for l in shapes):
            lens = shapes[0] if shapes else [0] * ncols
            cl = (ctypes.c_size_t * max(ncols, 1))(*lens)
            check(self._lib.h2v_verify_each(self._h, n, pa, pl, ia, ncols, cl, st))
            return list(st)[:n]
        # SingleStrategy proofs are independent: run every instance shape as its own call and put the statuses back in order
        out = [0] * n
        by_shape = {}
        for i, l in enumerate(shapes):
            by_shape.setdefault(tuple(l), []).append(i)
        for l, idx in by_shape.items():
            sub = self.verify_each([proofs[i] for i in idx], [instances[i] for i in idx])
            for i, v in zip(idx, sub):
                out[i] = v
        return out

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

    def __init__(self, params, rand=None, device=0, circuit_instances=1):
        super().__init__(params)
        self.rand, self.device, self.circuit_instances = rand, device, circuit_instances
        self.seed = None
        self.left_xy = self.right_xy = None   # the evaluated channels after finalize() (single-VK accumulations)

    @classmethod
    def with_accumulator(cls, params, left, right, rand=None, device=0, circuit_instances=1):
        """AccumulatorStrategy::with(msm_accumulator) (kzg/strategy.rs:75-78): start from an existing DualMSM — left / right are
        (scalars, bases) term lists as MSMKZG holds them.  A finished accumulation is resumed with left = ([1], [left_xy]),
        right = ([1], [right_xy])."""
        s = cls(params, rand=rand, device=device, circuit_instances=circuit_instances)
        s.seed = (left, right)
        return s

    def finalize(self) -> bool:
        """One pairing for everything that was accumulated.  verify_proof takes a VK per call and one strategy may accumulate
        proofs of DIFFERENT VKs over the same params (kzg/strategy.rs:125-140 only ever sees MSMs): proofs are grouped by VK,
        every VK gets its own context and accumulator pair (no pairing), the draws are indexed by call order over ALL queued
        proofs (proof i is scaled by the product of the draws of all later proofs, whatever their VK), and the accumulator
        records are folded by h2v_fold_check into the single pairing."""
        if not self._items:
            return True  # empty DualMSM: both channels are the identity, e(0,..)e(0,..) == 1
        n = len(self._items)
        rand = self.rand
        if rand is not None and len(rand) != n:
            raise ValueError(f"rand must hold one scalar per accumulated proof ({n}), got {len(rand)}")
        groups = {}
        for i, (vk, _, _) in enumerate(self._items):
            groups.setdefault((vk.data, int(vk.format)), []).append(i)
        if len(groups) == 1:
            ctx = Context(self.params, self._items[0][0], self.device, circuit_instances=self.circuit_instances)
            try:
                ok, _, self.left_xy, self.right_xy = ctx.verify_batch([p for _, _, p in self._items], [i for _, i, _ in self._items], rand, seed=self.seed)
                return ok
            finally:
                ctx.close()
        if self.seed is not None:
            raise ValueError("a seeded accumulation takes proofs of one VerifyingKey")
        return self._finalize_mixed(groups, rand)

    def _finalize_mixed(self, groups, rand):
        import torch  # device memory for the gathered records (plumbing only)
        lib = _lib.load_library()
        n = len(self._items)
        if rand is None:
            rand = [int.from_bytes(os.urandom(64), "little") % _FR_MODULUS for _ in range(n)]
        rb = [_scalar32(r) for r in rand]
        # A VK's proofs are not contiguous in call order, so the library's "tail" convention (multiplier = product of the later
        # draws of the SAME upload) is fed per proof: proof i is uploaded as a one-proof shard whose tail is the draws of
        # proofs (i, n) of the whole accumulated sequence.  Mixed-VK accumulation is a rare path; clarity over speed.
        records = torch.zeros(n * distributed.ACC_BYTES, dtype=torch.uint8, device=f"cuda:{self.device}")
        ctxs, ok_all = {}, True
        try:
            for key, idx in groups.items():
                ctx = ctxs[key] = Context(self.params, self._items[idx[0]][0], self.device, circuit_instances=self.circuit_instances)
                for i in idx:
                    _, inst, proof = self._items[i]
                    flat, lens = _flatten_instances(inst)
                    b = Batch(ctx, 1, max(sum(lens), 1))
                    try:
                        b.upload(proof, len(proof), flat, lens, b"".join(rb[i:]))
                        b.launch(with_pairing=False)
                        b.export_accumulators(records.data_ptr() + i * distributed.ACC_BYTES)
                        _, st, _, _ = b.finish()
                        ok_all = ok_all and st == [0]
                    finally:
                        b.close()
            ok = ctypes.c_int(0)
            any_ctx = next(iter(ctxs.values()))
            check(lib.h2v_fold_check(any_ctx._h, ctypes.c_void_p(records.data_ptr()), n, ctypes.byref(ok), None, None))
            return bool(ok.value) and ok_all
        finally:
            for c in ctxs.values():
                c.close()


class SingleStrategy(_Strategy):
    """poly/kzg/strategy.rs:83-102,164-176: one pairing per proof, checked inside verify_proof."""

    def __init__(self, params, device=0, circuit_instances=1):
        super().__init__(params)
        self.device, self.circuit_instances = device, circuit_instances


def verify_proof(params: ParamsKZG, vk: VerifyingKey, strategy, instances, proof: bytes):
    """lib.rs:33-49.  `instances` = one circuit instance: list of columns — or, for a strategy created with circuit_instances = M,
    the M x columns of the M instances that share the transcript, instance by instance (the reference's `&[&[&[Fr]]]` flattened).
    With SingleStrategy returns None or raises H2VError(code = PlonkError); with AccumulatorStrategy returns the strategy
    (Output = Self)."""
    if isinstance(strategy, SingleStrategy):
        ctx = Context(params, vk, strategy.device, circuit_instances=strategy.circuit_instances)
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

    STAGES = ("decompress", "transcript", "fr_program", "fold", "msm", "pairing", "msm_accumulate")   # the last one lies inside "msm"

    def __init__(self, ctx: Context, max_proofs: int, max_instance_values: int = 0, stream=None, groups: int = 1):
        self.ctx, self._lib = ctx, ctx._lib
        self._h = ctypes.c_void_p()
        check(self._lib.h2v_batch_create(ctx._h, max_proofs, max_instance_values, ctypes.byref(self._h)))
        if not hasattr(ctx, "_batches"):
            import weakref
            ctx._batches = weakref.WeakSet()
        ctx._batches.add(self)
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
        # the C side reads n * proof_len, n * sum(col_lens) * 32 and n_tail * 32 bytes: the buffers must hold exactly that
        if proof_len and len(proofs_flat) != n * proof_len:
            raise ValueError("proofs_flat is not a whole number of proofs")
        if len(instances_flat) != n * sum(col_lens) * 32:
            raise ValueError(f"instances_flat must be n * sum(col_lens) * 32 = {n * sum(col_lens) * 32} bytes, got {len(instances_flat)}")
        if rand_tail is not None and len(rand_tail) % 32:
            raise ValueError("rand_tail is not a whole number of 32-byte scalars")
        cl = (ctypes.c_size_t * max(len(col_lens), 1))(*col_lens)
        nt = len(rand_tail) // 32 if rand_tail is not None else 0
        check(self._lib.h2v_batch_upload(self._h, n, proofs_flat, proof_len, instances_flat, len(col_lens), cl, rand_tail, nt))
        self.n = n

    def launch(self, with_pairing=True):
        check(self._lib.h2v_batch_launch(self._h, 1 if with_pairing else 0))

    def upload_launch(self, proofs_flat: bytes, proof_len: int, instances_flat: bytes, col_lens, rand_tail=None, with_pairing=True):
        """upload() + launch() with the host -> device copy hidden behind the point decompression (h2v_batch_upload_launch)."""
        n = len(proofs_flat) // proof_len if proof_len else 0
        if proof_len and len(proofs_flat) != n * proof_len:
            raise ValueError("proofs_flat is not a whole number of proofs")
        if len(instances_flat) != n * sum(col_lens) * 32:
            raise ValueError(f"instances_flat must be n * sum(col_lens) * 32 = {n * sum(col_lens) * 32} bytes, got {len(instances_flat)}")
        if rand_tail is not None and len(rand_tail) % 32:
            raise ValueError("rand_tail is not a whole number of 32-byte scalars")
        cl = (ctypes.c_size_t * max(len(col_lens), 1))(*col_lens)
        nt = len(rand_tail) // 32 if rand_tail is not None else 0
        check(self._lib.h2v_batch_upload_launch(self._h, n, proofs_flat, proof_len, instances_flat, len(col_lens), cl, rand_tail, nt, 1 if with_pairing else 0))
        self.n = n

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
        return bool(ok.value), memoryview(st).cast('B').cast('i').tolist()[:self.n], left.raw, right.raw

    def finish_groups(self, raw_statuses=False):
        """-> (group_ok[groups], statuses, left_xy[groups], right_xy[groups]).
        `raw_statuses`: the per-proof statuses as the bytes of the C array (n little-endian int32; all zero <=> every proof OK) instead of
        a list — turning 20 480 statuses into Python ints costs 0.15 ms, 5 % of the launch they come from."""
        g = self.groups
        st = (ctypes.c_int * max(self.n, 1))()
        ok = (ctypes.c_int * g)()
        left, right = ctypes.create_string_buffer(64 * g), ctypes.create_string_buffer(64 * g)
        check(self._lib.h2v_batch_finish_groups(self._h, st, ok, left, right, g))
        lr, rr = left.raw, right.raw
        statuses = bytes(memoryview(st).cast('B')[:4 * self.n]) if raw_statuses else memoryview(st).cast('B').cast('i').tolist()[:self.n]
        return ([bool(v) for v in ok], statuses, [lr[64 * i:64 * i + 64] for i in range(g)], [rr[64 * i:64 * i + 64] for i in range(g)])

    PROFILE_KERNEL = 3   # H2V_PROFILE_KERNEL: the dominant kernel's own timestamps only

    def set_profiling(self, on=True):
        """True / 1: an event between the stages and around the dominant kernel (each a barrier packet on the stream);
        Batch.PROFILE_KERNEL: the dominant kernel's own timestamps only (no extra packet); False / 0: off"""
        check(self._lib.h2v_batch_set_profiling(self._h, int(on)))

    def timings_ms(self):
        arr = (ctypes.c_float * len(self.STAGES))()
        k = self._lib.h2v_batch_timings(self._h, arr, len(self.STAGES))
        return dict(zip(self.STAGES, list(arr)[:k]))
