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


def _compact(lo, hi, step):
    """a synthetic compact export of inputs lo..hi-1 whose result counts depend on the step"""
    counts = [(q * 7 + step * 3) % 6 for q in range(lo, hi)]
    off = np.zeros(hi - lo + 1, dtype="<u4")
    off[1:] = np.cumsum(counts)
    rows = np.zeros(int(off[-1]), dtype=shard.TOPK_DTYPE)
    k = 0
    for q, c in zip(range(lo, hi), counts):
        for i in range(c):
            rows[k] = (q * 100 + i + step, 1.0 / (i + 1 + step), 1.0 - i / 64.0)
            k += 1
    head = off.tobytes()
    head += b"\0" * (shard.compact_offsets_bytes(hi - lo) - len(head))
    return head + rows.tobytes()


def _compact_worker(rank, world, port, n, steps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.shard_range(n, rank, world)
    per = max(shard.shard_range(n, r, world)[1] - shard.shard_range(n, r, world)[0] for r in range(world))
    g = shard.CompactGather(shard.compact_capacity(per, 5), "cpu", rank, world)
    seen = {}

    def collect(slot, step):
        if rank == 0 and step >= 0:
            parts = g.result(slot)
            seen[step] = [shard.decode_compact(parts[r], shard.shard_range(n, r, world)[1] - shard.shard_range(n, r, world)[0])
                          for r in range(world)]
    for step in range(steps):
        slot = step & 1
        buf = g.acquire(slot)          # the transfer of step - 2 has finished: its result is complete
        collect(slot, step - 2)
        payload = _compact(lo, hi, step)
        buf[:len(payload)] = torch.from_numpy(np.frombuffer(payload, dtype=np.uint8).copy())
        g.submit(slot, len(payload))
    g.flush()
    for step in (steps - 2, steps - 1):
        collect(step & 1, step)
    if rank == 0:
        q.put(seen)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,steps", [(2, 11, 5), (3, 10, 4), (1, 4, 3)])
def test_compact_gather_pipeline(world, n, steps):
    """the variable-size gather of compact exports: sizes one step ahead of the payloads, two rotating buffers"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_compact_worker, args=(r, world, port, n, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    seen = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(seen) == list(range(steps))
    for step in range(steps):
        for r in range(world):
            lo, hi = shard.shard_range(n, r, world)
            assert seen[step][r] == shard.decode_compact(_compact(lo, hi, step), hi - lo), (step, r)


def test_shard_range_partition():
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            r = [shard.shard_range(n, i, w) for i in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
