"""The sharded job end to end on real engine output: two one-GPU ranks (both on device 0 here, gloo
for the exchange: the pool's boxes have one GPU) shard the queries, run the device pipeline, export their compact top-k
records and rank 0 receives them through analiticcl_amd.shard.CompactGather; the decoded parts must equal the single-
process results of the same queries.  The NCCL path itself (bench.py --gpus N) needs a multi-GPU node."""
import os
import socket

import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nq, steps, q):
    import torch
    import torch.distributed as dist

    import analiticcl_amd as A
    from analiticcl_amd import shard, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = synth.materialize_golden(f"/tmp/anx_dist_test_{os.getuid()}_{rank}")
    g = A.VariantModel(d["alphabet"], A.Weights(), device=0)
    g.read_lexicon(d["eng"])
    g.build()
    queries = synth.make_queries(synth.load_lexicon_words(d["eng"]), nq, max_len=16, seed=21) + ["", "zzzzqqqq"]
    p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10, score_threshold=0.25, cutoff_threshold=2.0)
    lo, hi = shard.shard_range(len(queries), rank, world)
    mine = queries[lo:hi]
    b = g.encode_batch(mine, p)
    cap = shard.compact_capacity(len(mine) + 1, 16)
    dev = torch.empty(cap, dtype=torch.uint8, device="cuda:0")
    gather = shard.CompactGather(cap, "cpu", rank, world)
    seen = {}
    for step in range(steps):
        slot = step & 1
        buf = gather.acquire(slot)
        if rank == 0 and step >= 2:
            seen[step - 2] = [shard.decode_compact(part, shard.shard_range(len(queries), r, world)[1] - shard.shard_range(len(queries), r, world)[0])
                              for r, part in enumerate(gather.result(slot))]
        b.run()
        used = b.export_compact(dev.data_ptr(), cap)
        torch.cuda.synchronize()
        buf[:used].copy_(dev[:used])
        gather.submit(slot, used)
    gather.flush()
    if rank == 0:
        for step in (steps - 2, steps - 1):
            seen[step] = [shard.decode_compact(part, shard.shard_range(len(queries), r, world)[1] - shard.shard_range(len(queries), r, world)[0])
                          for r, part in enumerate(gather.result(step & 1))]
        whole = g.find_variants_ids(queries, p)
        q.put((seen, whole))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_export_gather():
    import torch.multiprocessing as mp
    world, nq, steps = 2, 3000, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nq, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    seen, whole = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(seen) == list(range(steps))
    for step in range(steps):
        merged = [row for part in seen[step] for row in part]
        assert len(merged) == len(whole)
        for got, exp in zip(merged, whole):
            assert [(v, d) for v, d, _ in got] == [(v, d) for v, d, _ in exp]
            assert all(abs(a[2] - c[2]) < 1e-6 for a, c in zip(got, exp))


def test_bench_rccl_path_world1():
    """What a one-GPU box can exercise of bench.py's RCCL path: launched by torch.distributed.run with ONE rank, the process
    group is initialised with backend nccl (= RCCL), the size exchange of the compact gather, the all-reduces and barriers run
    over it, and rank 0's gathered export decodes to the rows fetch() returns.  N > 1 needs a multi-GPU node (the driver's)."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(repo, "bench.py"), "--gpus", "1", "--force-gather", "--check-gather",
           "--steps", "3", "--warmup", "1", "--queries", "200000", "--cpu-sample", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=repo)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    js = json.loads(line)
    assert js["process_group"] == "nccl" and js["gather_error"] is None and js["gather_check"] == "ok"
    assert "RCCL gather" in js["config"]["parallelism"] and js["n_gpus"] == 1 and js["value"] > 0


def _bench(args, timeout=900):
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, cwd=repo, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_bench_single_process_replicas():
    """bench.py --single-process: ONE process, the C ABI's own split over replicas (here 3 on the one GPU): same JSON contract,
    the rows of the last shard equal a one-replica run of its inputs."""
    js = _bench(["--gpus", "3", "--single-process", "--replicas-on-one-gpu", "--queries", "150000", "--steps", "3", "--warmup", "1", "--cpu-sample", "2000"])
    assert js["n_gpus"] == 3 and js["shard_check"] == "ok" and js["value"] > 0 and js["scaling"] == "weak"
    # the length-partitioned split balances the replicas by COST (short queries are dearer): three shares that hold every input once
    assert [s[0] for s in js["shards"]] == [0, 0, 0] and sum(s[2] for s in js["shards"]) == 450000 and all(s[2] > 30000 for s in js["shards"])
    assert js["roofline"]["frac"] > 0 and js["cpu_baseline"]["value"] > 0 and "anx_model_to_devices" in js["config"]["parallelism"]


def test_bench_n_ranks_on_one_gpu_gloo():
    """Dry run of the N-rank job on one GPU: 3 processes under torch.distributed.run, all on device 0, the index built once and
    loaded by the other ranks, compact records gathered over gloo; rank 0's view of EVERY rank's export equals that rank's fetch();
    roofline and cpu_baseline are reported at N > 1 too."""
    js = _bench(["--ranks-on-one-gpu", "3", "--backend", "gloo", "--check-gather", "--queries", "120000", "--steps", "3", "--warmup", "1", "--cpu-sample", "2000"])
    assert js["n_gpus"] == 3 and js["process_group"] == "gloo" and js["ranks_on_one_gpu"] is True
    assert js["gather_error"] is None and js["gather_check"] == "ok"
    assert js["queries_per_s"] > 0 and js["roofline"]["frac"] > 0 and js["cpu_baseline"]["value"] > 0
