"""world_size-2 (and 3) gloo test of the multi-GPU plumbing: query sharding + the single gather of fixed-stride
top-k records (the only exchange of the path).  No device compute here; the records are synthetic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from analiticcl_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _records(lo, hi, stride):
    a = np.zeros((hi - lo, stride), dtype=shard.TOPK_DTYPE)
    a["vocab_id"] = shard.EMPTY
    for q in range(lo, hi):
        n = q % (stride + 1)
        for i in range(n):
            a[q - lo, i] = (q * 100 + i, 1.0 / (i + 1), 1.0 - i / 64.0)
    return a


def _worker(rank, world, port, n, stride, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.shard_range(n, rank, world)
    local = torch.from_numpy(np.frombuffer(_records(lo, hi, stride).tobytes(), dtype=np.uint8).copy())
    out = shard.gather_topk(local, n, stride, rank, world)
    if rank == 0:
        q.put(shard.decode_topk(out, n, stride))
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 11), (3, 10)])
def test_shard_and_gather(world, n):
    stride = 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, stride, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    exp = shard.decode_topk(np.frombuffer(_records(0, n, stride).tobytes(), dtype=np.uint8).tobytes(), n, stride)
    assert got == exp


def test_shard_range_partition():
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            r = [shard.shard_range(n, i, w) for i in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
