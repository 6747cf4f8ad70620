"""Sharded verification of one batch across the GPUs of a node (one process per GPU).

The reference has no communication layer.  Proofs are independent until the final
pairing, so the batch is cut into contiguous shards; every rank runs the whole per-proof pipeline
and its two pooled MSMs, and the only exchange is an all-gather of one 1312-byte record per rank and group (the two accumulators in
the six pieces the launch leaves them in, 108 bytes each, + the shard's failed-proof count) — RCCL has no user-defined reduction, and
a group addition is not a numeric sum — followed by a piece-wise fold and ONE pairing over the pieces (DualMSM::add_msm + check,
poly/kzg/msm.rs:178-203).

Multipliers: proof i of the whole batch is scaled by the product of the Fr::random draws of all
later proofs (kzg/strategy.rs:129, msm.rs:173-176), indexed globally, so the result does not
depend on the number of ranks.
"""


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
