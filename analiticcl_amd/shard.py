"""Query sharding across one-process-per-GPU ranks and the single result gather.

Queries are independent (the model is immutable during find_variants, /root/reference/src/lib.rs:972), so rank r
of W takes the contiguous slice [r*n/W, (r+1)*n/W) and there is no data-path collective; the only exchange is the
gather of fixed-stride anx_topk_record rows (include/anx.h) to rank 0.  Works with any torch.distributed backend
(nccl = RCCL on the GPU box, gloo in the CPU tests).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

TOPK_DTYPE = np.dtype([("vocab_id", "<u4"), ("freq_score", "<f4"), ("dist_score", "<f8")])  # anx_topk_record
EMPTY = 0xFFFFFFFF


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice of n items owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_topk(local, n_total: int, stride: int, rank: int, world: int, dst: int = 0):
    """local: uint8 tensor holding this rank's (hi-lo)*stride records.  Returns on dst a uint8 tensor with all
    n_total*stride records in global query order, else None.  Slices are padded to equal size for the gather."""
    import torch
    import torch.distributed as dist

    rec = TOPK_DTYPE.itemsize
    per = max(shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world))
    pad = torch.empty(per * stride * rec, dtype=torch.uint8, device=local.device)
    pad[: local.numel()] = local
    if world == 1:
        return local
    out = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, out, dst=dst)
    if rank != dst:
        return None
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        parts.append(out[r][: (hi - lo) * stride * rec])
    return torch.cat(parts)


def decode_topk(buf, n: int, stride: int) -> List[List[Tuple[int, float, float]]]:
    """uint8 tensor / bytes -> per query [(vocab_id, dist_score, freq_score)] (empty slots dropped)."""
    raw = bytes(buf.cpu().numpy().tobytes()) if hasattr(buf, "cpu") else bytes(buf)
    a = np.frombuffer(raw, dtype=TOPK_DTYPE).reshape(n, stride)
    return [[(int(r["vocab_id"]), float(r["dist_score"]), float(r["freq_score"])) for r in row if r["vocab_id"] != EMPTY]
            for row in a]


# -- the gather of compact exports (anx_batch_export_compact): variable size, pipelined ---------------------------
def compact_offsets_bytes(n: int) -> int:
    return ((n + 1) * 4 + 15) & ~15


def compact_capacity(n: int, stride: int) -> int:
    """bytes that always suffice for n inputs with at most `stride` records each"""
    return compact_offsets_bytes(n) + n * stride * TOPK_DTYPE.itemsize


def decode_compact(buf, n: int) -> List[List[Tuple[int, float, float]]]:
    """compact export (uint8 tensor / bytes) -> per input [(vocab_id, dist_score, freq_score)]"""
    raw = bytes(buf.cpu().numpy().tobytes()) if hasattr(buf, "cpu") else bytes(buf)
    off = np.frombuffer(raw, dtype="<u4", count=n + 1)
    rows = np.frombuffer(raw, dtype=TOPK_DTYPE, count=int(off[n]), offset=compact_offsets_bytes(n))
    return [[(int(r["vocab_id"]), float(r["dist_score"]), float(r["freq_score"])) for r in rows[off[i]:off[i + 1]]]
            for i in range(n)]


class CompactGather:
    """Rank `dst` collects every rank's compact export, moving only the bytes in use.

    The sizes differ per rank and step, so each step first gathers one int64 per rank (asynchronously) and the payload
    follows one step later, when the sizes have long arrived: acquire(slot) -> export into the buffer ->
    submit(slot, used) posts the size exchange of this step and the point-to-point payload transfers of the previous
    one, which then overlap the next step's kernels.  `depth` buffers rotate (slot = step % depth); flush() posts and
    waits for what is still outstanding.  On `dst`, result(slot) lists the received exports per rank.
    Any torch.distributed backend: nccl (= RCCL over xGMI) on the GPUs, gloo in the CPU tests."""

    def __init__(self, capacity: int, device, rank: int, world: int, dst: int = 0, depth: int = 2):
        import torch

        self.rank, self.world, self.dst, self.depth = rank, world, dst, depth
        self.send = [torch.empty(capacity, dtype=torch.uint8, device=device) for _ in range(depth)]
        self.size_dev = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(depth)]
        self.recv = self.sizes = None
        # with an initialised process group the size exchange runs at world 1 too (one rank gathering from itself): that is
        # what a one-GPU box can exercise of the RCCL path
        import torch.distributed as dist
        self.collective = world > 1 or (dist.is_available() and dist.is_initialized())
        if rank == dst and self.collective:
            self.recv = [[torch.empty(capacity, dtype=torch.uint8, device=device) if r != dst else None
                          for r in range(world)] for _ in range(depth)]
            self.sizes = [[torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)] for _ in range(depth)]
        self.used = [0] * depth
        self.size_work = [None] * depth   # size exchange posted, payload not yet
        self.data_work = [None] * depth   # payload transfers in flight
        self.got = [None] * depth         # on dst: bytes received per rank
        self.order: List[int] = []        # slots whose payload still has to be posted, oldest first

    def acquire(self, slot: int):
        """the export buffer of `slot`, once the transfer that last used it has finished"""
        if self.size_work[slot] is not None:
            self._post_payload(slot)
        if self.data_work[slot] is not None:
            for w in self.data_work[slot]:
                w.wait()
            self.data_work[slot] = None
        return self.send[slot]

    def submit(self, slot: int, used: int) -> None:
        import torch.distributed as dist

        self.used[slot] = used
        if not self.collective:
            self.got[slot] = [used]
            return
        while self.order:  # payloads of earlier steps: their sizes were exchanged while this step computed
            self._post_payload(self.order[0])
        self.size_dev[slot].fill_(used)
        self.size_work[slot] = dist.gather(self.size_dev[slot], self.sizes[slot] if self.rank == self.dst else None,
                                           dst=self.dst, async_op=True)
        self.order.append(slot)

    def _post_payload(self, slot: int) -> None:
        import torch.distributed as dist

        self.size_work[slot].wait()
        self.size_work[slot] = None
        self.order.remove(slot)
        ops = []
        if self.rank == self.dst:
            sizes = [int(t.item()) for t in self.sizes[slot]]
            self.got[slot] = sizes
            for r in range(self.world):
                if r != self.dst and sizes[r] > 0:
                    ops.append(dist.P2POp(dist.irecv, self.recv[slot][r][:sizes[r]], r))
        elif self.used[slot] > 0:
            ops.append(dist.P2POp(dist.isend, self.send[slot][:self.used[slot]], self.dst))
        self.data_work[slot] = dist.batch_isend_irecv(ops) if ops else None

    def flush(self) -> None:
        while self.order:
            self._post_payload(self.order[0])
        for slot in range(self.depth):
            if self.data_work[slot] is not None:
                for w in self.data_work[slot]:
                    w.wait()
                self.data_work[slot] = None

    def result(self, slot: int):
        """on dst, after the slot's transfer finished (acquire / flush): the export of every rank (views)"""
        if self.rank != self.dst:
            return None
        return [(self.send[slot] if r == self.dst else self.recv[slot][r])[:self.got[slot][r]] for r in range(self.world)]
