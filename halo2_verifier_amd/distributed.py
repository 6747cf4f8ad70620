"""Sharded verification of one batch across the GPUs of a node (one process per GPU).

The reference has no communication layer.  Proofs are independent until the final pairing, so the batch is cut
into contiguous shards; every rank runs the whole per-proof pipeline and its two pooled MSMs, and the only exchange is an
all-gather of one 1312-byte record per rank and group (the two accumulators in the six pieces the launch leaves them in, 108 bytes
each, + the shard's failed-proof count) — RCCL has no user-defined reduction, and a group addition is not a numeric sum — followed
by a piece-wise fold and ONE pairing over the pieces (DualMSM::add_msm + check, poly/kzg/msm.rs:173-203).

Multipliers: proof i of the whole batch is scaled by the product of the Fr::random draws of all later proofs
(kzg/strategy.rs:129, msm.rs:173-176), indexed globally, so the result does not depend on the number of ranks: a rank uploads the
draws from its first proof to the END of the batch (its "tail").

Entry points
  verify_batch_sharded(ctx, proofs, instances, rand=None, group=None)
      N x verify_proof on ONE AccumulatorStrategy + finalize(), the proofs sharded over the ranks of a torch.distributed group.
      Every rank passes the whole batch and gets the same (ok, statuses, left_xy, right_xy).
  verify_batch_sharded_local(ctx, proofs, instances, rand, world)
      the same computation with the `world` shards run one after the other on ONE GPU (no process group): sharding invariance
      at full size on a single device, and hosts with one GPU.
  ShardedBatch
      the staged form both are built on (and bench.py pipelines): upload the shard once, launch() = shard pipeline -> export ->
      all-gather -> fold -> one pairing, asynchronous on the batch's stream; finish() fetches the verdict.
"""
import os

_FR_MODULUS = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001


def shard_bounds(total: int, world_size: int, rank: int):
    """Contiguous shard [lo, hi) of `total` proofs for `rank`."""
    base, rem = divmod(total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# include/h2v.h H2V_ACC_RECORD_BYTES: [u32 failed proofs of the shard][u32 parts][u32 shift][u32 0] + 2 x 6 Jacobian G1 pieces
# (3 coordinates x 9 limbs x 4 B, the library's Montgomery limb layout).  The count travels with the points so that every rank
# clears the verdict when ANY shard rejected a proof (a failed proof contributes nothing to its shard's accumulators).
ACC_BYTES = 16 + 2 * 6 * 108


def gather_accumulators(local_acc, world_size, group=None):
    """all_gather of the per-rank accumulator bytes (torch uint8 tensor of groups * ACC_BYTES, as Batch.export_accumulators
    writes them) -> tensor [world_size * groups * ACC_BYTES], the layout Batch.fold_check_enqueue folds group by group.
    Works with the nccl (= RCCL) backend on GPU tensors and with gloo on CPU tensors (tests)."""
    import torch
    import torch.distributed as dist
    out = torch.empty(world_size * local_acc.numel(), dtype=torch.uint8, device=local_acc.device)
    if world_size == 1:
        out.copy_(local_acc)
        return out
    dist.all_gather_into_tensor(out, local_acc, group=group)
    return out


def tail_for_shard(rand_all: bytes, lo: int):
    """The draws this shard needs: those of proofs [lo, total)."""
    return rand_all[32 * lo:]


def draw_scalars(n: int) -> bytes:
    """n x Fr::random(OsRng) as AccumulatorStrategy::process draws them (kzg/strategy.rs:129): 64 OS-random bytes reduced mod r,
    32 little-endian canonical bytes each."""
    try:   # the library's own generator (h2v_random_scalars: what rand32 = NULL draws inside the one-shot entry points)
        import ctypes
        from . import _lib
        buf = ctypes.create_string_buffer(max(32 * n, 1))
        _lib.check(_lib.load_library().h2v_random_scalars(buf, n))
        return buf.raw[: 32 * n]
    except _lib.H2VError:   # library not built (orchestration tests without the product): the same distribution from os.urandom
        return b"".join((int.from_bytes(os.urandom(64), "little") % _FR_MODULUS).to_bytes(32, "little") for _ in range(n))


def _scalar_bytes(rand) -> bytes:
    out = bytearray()
    for r in rand:
        if isinstance(r, (bytes, bytearray, memoryview)):
            if len(r) != 32:
                raise ValueError("a draw given as bytes must be exactly 32 bytes")
            out += bytes(r)
        else:
            r = int(r)
            if r < 0 or r >> 256:
                raise ValueError("a draw given as an integer must be in [0, 2^256)")
            out += r.to_bytes(32, "little")
    return bytes(out)


def _group_info(group):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1, None
    return dist.get_rank(group), dist.get_world_size(group), dist.get_backend(group)


def common_draws(n: int, rand, group=None, device=None) -> bytes:
    """The ONE stream of Fr::random draws every rank of a sharded batch must use: `rand` (n ints / 32-byte strings, the same on
    every rank) or, when None, drawn by rank 0 from the OS and broadcast.  -> n * 32 bytes."""
    rank, world, backend = _group_info(group)
    if rand is not None:
        if len(rand) != n:
            raise ValueError(f"rand must hold one scalar per proof ({n}), got {len(rand)}")
        return _scalar_bytes(rand)
    if world == 1:
        return draw_scalars(n)
    import torch
    import torch.distributed as dist
    dev = device if (backend == "nccl" and device is not None) else "cpu"
    buf = torch.zeros(max(32 * n, 1), dtype=torch.uint8, device=dev)
    if rank == 0 and n:
        buf[: 32 * n] = torch.frombuffer(bytearray(draw_scalars(n)), dtype=torch.uint8).to(dev)
    dist.broadcast(buf, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return bytes(buf[: 32 * n].cpu().numpy().tobytes())


class ShardedBatch:
    """One rank's share of a sharded batch (or of `groups` independent sharded batches that travel in one launch), resident on
    its GPU.  launch() enqueues, on the batch's stream and without host synchronisation: the shard's pipeline up to its two
    accumulators -> their records -> all-gather over the group -> fold -> ONE pairing per group; finish() returns what every rank
    agrees on.  With a world of one the launch simply ends in its own pairing.

    stream: a torch.cuda.Stream (or a raw HIP stream handle) the batch runs on — collectives are issued under the same stream, so MSM,
    all-gather and pairing are ordered without host synchronisation; None: a stream of the batch's own.
    batch_factory(ctx, max_proofs, max_instance_values, stream, groups) builds the object that runs a shard; the default is the
    HIP batch (verifier.Batch) — there is no CPU path in the product (the non-GPU test suite injects a stand-in to exercise the
    orchestration over gloo)."""

    def __init__(self, ctx, max_proofs: int, max_instance_values: int = 0, groups: int = 1, group=None, stream=None, device=None,
                 batch_factory=None, world_size=None, always_exchange=False):
        import torch
        self.rank, self.world, self.backend = _group_info(group)
        if world_size is not None:        # sequential simulation of `world_size` ranks (verify_batch_sharded_local): no process group
            self.world = world_size
        self.group, self.groups = group, groups
        self.always_exchange = always_exchange   # a world of one still runs export -> gather -> fold (what sharding costs besides the collective)
        if batch_factory is None:
            from .verifier import Batch
            batch_factory = lambda c, n, mi, st, g: Batch(c, n, mi, stream=st, groups=g)
            if device is None:
                device = f"cuda:{ctx.device}"
        self.device = torch.device(device if device is not None else "cpu")
        self._torch_stream = None
        if stream is not None and hasattr(stream, "cuda_stream"):   # a torch.cuda.Stream: collectives are issued under it as it is
            self._torch_stream = stream
            stream = stream.cuda_stream
        if stream is None and self.device.type == "cuda" and self.world > 1 and world_size is None:
            # the collective must be ordered with the batch's kernels: run both on one torch stream
            self._torch_stream = torch.cuda.Stream(device=self.device)
            stream = self._torch_stream.cuda_stream
        self._stream_handle = stream
        self.batch = batch_factory(ctx, max(max_proofs, 1), max_instance_values, stream, groups)
        self.records = torch.zeros(ACC_BYTES * groups, dtype=torch.uint8, device=self.device)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)   # (allocated and zeroed on torch's current stream, written on the batch's)
        self.gathered = None

    def close(self):
        if self.batch is not None:
            self.batch.close()
            self.batch = None

    def upload(self, proofs_flat: bytes, proof_len: int, instances_flat: bytes, col_lens, rand_tail: bytes):
        """This rank's shard (of every group: group-major) and, per group, the draws from its first proof to the end of the
        group's whole batch."""
        self.batch.upload(proofs_flat, proof_len, instances_flat, col_lens, rand_tail)

    def launch_shard(self):
        """Shard pipeline without a pairing + export of the accumulator records -> the local record tensor (stream-ordered)."""
        self.batch.launch(with_pairing=False)
        self.batch.export_accumulators(self.records.data_ptr())
        return self.records

    def fold(self, gathered, n_records: int):
        """Fold `n_records` gathered record sets ([rank][group]) and enqueue the one pairing per group."""
        self.gathered = gathered      # kept alive until finish()
        self.batch.fold_check_enqueue(gathered.data_ptr(), n_records)

    def upload_launch(self, proofs_flat: bytes, proof_len: int, instances_flat: bytes, col_lens, rand_tail: bytes):
        """upload() + launch() with the host -> device copy of the shard hidden behind its point decompression
        (h2v_batch_upload_launch): for shards that arrive from the host for every batch."""
        if self.world == 1 and not self.always_exchange:
            self.batch.upload_launch(proofs_flat, proof_len, instances_flat, col_lens, rand_tail, with_pairing=True)
            return
        import torch
        ctxm = torch.cuda.stream(self._external_stream()) if (self.device.type == "cuda" and self._stream_handle is not None) else _NullCtx()
        with ctxm:
            self.batch.upload_launch(proofs_flat, proof_len, instances_flat, col_lens, rand_tail, with_pairing=False)
            self.batch.export_accumulators(self.records.data_ptr())
            self.fold(gather_accumulators(self.records, self.world, self.group), self.world)

    def launch(self):
        if self.world == 1 and not self.always_exchange:
            self.batch.launch(with_pairing=True)
            return
        import torch
        ctxm = torch.cuda.stream(self._external_stream()) if (self.device.type == "cuda" and self._stream_handle is not None) else _NullCtx()
        with ctxm:
            local = self.launch_shard()
            self.fold(gather_accumulators(local, self.world, self.group), self.world)

    def _external_stream(self):
        import torch
        if self._torch_stream is not None:
            return self._torch_stream
        return torch.cuda.ExternalStream(self._stream_handle, device=self.device)

    def finish(self, raw_statuses=False):
        """-> (group_ok[groups], local statuses, left_xy[groups], right_xy[groups]); the verdict and the accumulators are those of
        the WHOLE sharded batch (identical on every rank), the statuses are this rank's proofs'."""
        return self.batch.finish_groups(raw_statuses=raw_statuses)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _flatten(proofs, instances):
    """-> (proofs_flat, proof_len, instances_flat, col_lens); one instance shape per sharded batch (the staged interface)."""
    from .verifier import _flatten_instances
    if len(instances) != len(proofs):
        raise ValueError(f"{len(proofs)} proofs but {len(instances)} instance lists")
    plen = len(proofs[0]) if proofs else 0
    if any(len(p) != plen for p in proofs):
        raise ValueError("a sharded batch takes proofs of one length (one VerifyingKey)")
    flats, lens0 = [], None
    for inst in instances:
        f, lens = _flatten_instances(inst)
        if lens0 is None:
            lens0 = lens
        elif lens != lens0:
            raise ValueError("a sharded batch takes one instance shape (use Context.verify_batch for mixed shapes)")
        flats.append(f)
    return b"".join(bytes(p) for p in proofs), plen, b"".join(flats), lens0


def _all_statuses(local, n, world, group, backend, device):
    """Per-proof statuses of the whole batch, in call order, on every rank (a second, 4 n-byte all-gather)."""
    if world == 1:
        return list(local)
    import torch
    import torch.distributed as dist
    per = (n + world - 1) // world
    dev = device if backend == "nccl" else "cpu"
    mine = torch.zeros(max(per, 1), dtype=torch.int32, device=dev)
    if local:
        mine[: len(local)] = torch.tensor(local, dtype=torch.int32, device=dev)
    out = torch.empty(world * max(per, 1), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(out, mine, group=group)
    out = out.cpu().tolist()
    res = []
    for r in range(world):
        lo, hi = shard_bounds(n, world, r)
        res += out[r * max(per, 1): r * max(per, 1) + (hi - lo)]
    return res


def verify_batch_sharded(ctx, proofs, instances, rand=None, group=None, batch_factory=None, device=None):
    """N x verify_proof under ONE AccumulatorStrategy followed by finalize() (lib.rs:33-425, kzg/strategy.rs:125-140) with the
    proofs sharded over the ranks of `group` (default: the world) — BASELINE.json configs 3 and 5.

    Every rank calls it with the SAME `proofs` / `instances` (the whole batch in call order; one instance shape) on its own
    Context (one per GPU).  rand: the n Fr::random draws in call order, the same on every rank — or None: rank 0 draws them from
    the OS and broadcasts.  Rank r verifies shard_bounds(n, world, r) with the draw tail of its global position, the 1312-byte
    accumulator records are all-gathered (RCCL under the nccl backend), folded, and ONE pairing closes the batch.
    -> (ok, statuses[n], left_xy, right_xy), identical on every rank and — for given draws — for every world size."""
    rank, world, backend = _group_info(group)
    n = len(proofs)
    if device is None and batch_factory is None:
        device = f"cuda:{ctx.device}"
    draws = common_draws(n, rand, group, device)
    lo, hi = shard_bounds(n, world, rank)
    flat, plen, iflat, lens = _flatten(proofs[lo:hi], instances[lo:hi])
    if lens is None:                       # an empty shard still takes part in the collectives
        _, _, _, lens = _flatten(proofs[:1], instances[:1]) if n else (b"", 0, b"", [0] * ctx.proof_shape()["n_instance_columns"])
    if not plen:
        plen = len(proofs[0]) if n else ctx.proof_shape()["proof_len"]
    sb = ShardedBatch(ctx, hi - lo, max(sum(lens), 1), group=group, device=device, batch_factory=batch_factory)
    try:
        sb.upload(flat, plen, iflat, lens, tail_for_shard(draws, lo))
        sb.launch()
        ok, st, left, right = sb.finish()
    finally:
        sb.close()
    statuses = _all_statuses(st, n, world, group, backend, sb.device)
    return bool(ok[0]), statuses, left[0], right[0]


def verify_batch_sharded_local(ctx, proofs, instances, rand, world: int, batch_factory=None, device=None):
    """The `world`-rank computation of verify_batch_sharded run shard after shard on ONE device, no process group: every shard is
    uploaded with the draw tail of its global position and launched without a pairing, the records are laid out as the all-gather
    would, shard 0 folds them and runs the one pairing.  -> (ok, statuses[n], left_xy, right_xy)."""
    import torch
    n = len(proofs)
    if rand is not None and len(rand) != n:
        raise ValueError(f"rand must hold one scalar per proof ({n}), got {len(rand)}")
    draws = _scalar_bytes(rand) if rand is not None else draw_scalars(n)
    if device is None and batch_factory is None:
        device = f"cuda:{ctx.device}"
    shards, statuses = [], []
    try:
        for r in range(world):
            lo, hi = shard_bounds(n, world, r)
            flat, plen, iflat, lens = _flatten(proofs[lo:hi], instances[lo:hi])
            if lens is None:
                _, _, _, lens = _flatten(proofs[:1], instances[:1])
                plen = len(proofs[0])
            sb = ShardedBatch(ctx, hi - lo, max(sum(lens), 1), device=device, batch_factory=batch_factory, world_size=world)
            shards.append(sb)
            sb.upload(flat, plen, iflat, lens, tail_for_shard(draws, lo))
            sb.launch_shard()
            _, st, _, _ = sb.finish()
            statuses += st
        gathered = torch.cat([sb.records for sb in shards])
        if gathered.device.type == "cuda":
            torch.cuda.synchronize(gathered.device)
        shards[0].fold(gathered, world)
        ok, _, left, right = shards[0].finish()
    finally:
        for sb in shards:
            sb.close()
    return bool(ok[0]), statuses, left[0], right[0]
