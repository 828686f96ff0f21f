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
