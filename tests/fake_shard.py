"""Test stand-in for verifier.Batch with the CPU oracle behind it (NOT part of the product: halo2_verifier_amd has no CPU path).

distributed.ShardedBatch / verify_batch_sharded take a `batch_factory`; the non-GPU suite passes this one so that the
orchestration — common draws, shard bounds, draw tails, the all-gather of opaque 1312-byte records in [rank][group] order, the
fold, the status gather — runs over gloo with two ranks in a container without a GPU.  The record layout here is the stand-in's
own ([failed][1][0][0] + left x|y + right x|y, affine): the real layout is covered by the GPU suite."""
import ctypes

import circuits
import oracle_lib
from circuits import R_MOD
from halo2_verifier_amd.distributed import ACC_BYTES


def make_factory(setup):
    return lambda ctx, n, mi, stream, groups: OracleBatch(setup, groups)


class OracleBatch:
    def __init__(self, setup, groups):
        assert groups == 1
        self.s, self.L = setup, oracle_lib.load()
        self.P, self.I, self.draws = [], [], []
        self.res = None

    def close(self):
        pass

    def upload(self, proofs_flat, proof_len, instances_flat, col_lens, rand_tail):
        n = len(proofs_flat) // proof_len if proof_len else 0
        per = sum(col_lens) * 32
        self.P = [proofs_flat[i * proof_len:(i + 1) * proof_len] for i in range(n)]
        self.I = []
        for i in range(n):
            blob, cols, off = instances_flat[i * per:(i + 1) * per], [], 0
            for l in col_lens:
                cols.append([blob[off + 32 * j: off + 32 * j + 32] for j in range(l)]); off += 32 * l
            self.I.append(cols)
        self.draws = [int.from_bytes(rand_tail[32 * i:32 * i + 32], "little") for i in range(len(rand_tail) // 32)]

    def launch(self, with_pairing=True):
        n = len(self.P)
        T = 1
        for d in self.draws[n:]:
            T = T * d % R_MOD
        if n:
            ok, st, left, right = circuits.oracle_verify_batch(self.s, self.P, self.I, self.draws[:n])
            left, right = oracle_lib.g1_msm(self.L, [T], [left]), oracle_lib.g1_msm(self.L, [T], [right])
        else:
            ok, st, left, right = True, [], bytes(64), bytes(64)
        if not with_pairing:
            ok = not any(st)
        self.res = (ok, st, left, right)

    def export_accumulators(self, ptr):
        ok, st, left, right = self.res
        rec = sum(1 for v in st if v).to_bytes(4, "little") + (1).to_bytes(4, "little") + bytes(8) + left + right
        rec += bytes(ACC_BYTES - len(rec))
        ctypes.memmove(ptr, rec, ACC_BYTES)

    def fold_check_enqueue(self, ptr, n_records):
        blob = ctypes.string_at(ptr, n_records * ACC_BYTES)
        recs = [blob[i * ACC_BYTES:(i + 1) * ACC_BYTES] for i in range(n_records)]
        failed = sum(int.from_bytes(r[:4], "little") for r in recs)
        left = oracle_lib.g1_msm(self.L, [1] * n_records, [r[16:80] for r in recs])
        right = oracle_lib.g1_msm(self.L, [1] * n_records, [r[80:144] for r in recs])
        ok = circuits.oracle_pairing_check(self.s, left, right) and failed == 0 and not any(self.res[1])
        self.res = (ok, self.res[1], left, right)

    def finish_groups(self, raw_statuses=False):
        ok, st, left, right = self.res
        return [ok], list(st), [left], [right]
