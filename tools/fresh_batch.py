#!/usr/bin/env python3
"""The fresh-batch step taken apart (round 6) and the small call (item 4).

  fresh_batch.py loop        the with_encode loop of bench.py: device encoder of step i + 1 under the run of step i
  fresh_batch.py encode      the encoder alone, batch after batch (each freed at once)
  fresh_batch.py run         first runs alone: batches encoded ahead, then run one after the other (async, waited a step later)
  fresh_batch.py rerun       the resident re-run of bench.py's `value` (two copies alternating)
  fresh_batch.py small       anx_find_variants_batch host-to-host for n = 1 .. 1 M

Every mode prints one JSON line.  Run a mode directly behind `rocprofv3 --kernel-trace -- python3 tools/fresh_batch.py <mode>`
and feed the database to tools/timeline.py for the per-kernel "alone / in the loop / queue delay" table.
"""
import ctypes as C
import json
import os
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def setup(nq=1_000_000, lexicon=os.environ.get("FB_LEX", "eng"), max_len=int(os.environ.get("FB_MAXLEN", "16"))):
    import torch

    import analiticcl_amd as A
    from analiticcl_amd import synth
    torch.cuda.set_device(0)
    paths = synth.materialize_golden(os.path.join(tempfile.gettempdir(), f"anx_bench_data_{os.getuid()}_0"))
    model = A.VariantModel(paths["alphabet"], A.Weights(), device=0)
    model.read_lexicon(paths[lexicon])
    model.build()
    words = synth.load_lexicon_words(paths[lexicon])
    queries = synth.make_queries(words, nq, max_len=max_len, seed=synth.SEED)
    params = A.SearchParameters(max_anagram_distance=3, max_edit_distance=int(os.environ.get("FB_D", "2")), max_matches=10, score_threshold=0.25, cutoff_threshold=2.0)
    return torch, A, model, queries, params, paths


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "loop"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    torch, A, model, queries, params, paths = setup()
    stream = torch.cuda.current_stream().cuda_stream
    packed = ("\0".join(queries) + "\0").encode("utf-8")
    dev_blob = torch.frombuffer(bytearray(packed), dtype=torch.uint8).cuda()
    out = {"mode": mode, "steps": steps}
    if mode == "loop":
        live = []

        def enc_step():
            b = model.encode_packed_device(dev_blob.data_ptr(), dev_blob.numel(), len(queries), params)
            b.run_async(stream)
            live.append(b)
            if len(live) > 1:
                o = live.pop(0)
                o.wait()
                o.free()

        def drain():
            while live:
                o = live.pop(0)
                o.wait()
                o.free()
            torch.cuda.synchronize()
        for _ in range(4):
            enc_step()
        drain()
        t = time.perf_counter()
        for _ in range(steps):
            enc_step()
        drain()
        out["ms_per_step"] = (time.perf_counter() - t) / steps * 1e3
    elif mode == "encode":
        for _ in range(3):
            model.encode_packed_device(dev_blob.data_ptr(), dev_blob.numel(), len(queries), params).free()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            model.encode_packed_device(dev_blob.data_ptr(), dev_blob.numel(), len(queries), params).free()
        torch.cuda.synchronize()
        out["ms_per_step"] = (time.perf_counter() - t) / steps * 1e3
    elif mode == "run":
        def one_pass(n):
            bs = [model.encode_packed_device(dev_blob.data_ptr(), dev_blob.numel(), len(queries), params) for _ in range(n)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            prev = None
            for b in bs:
                b.run_async(stream)
                if prev is not None:
                    prev.wait()
                prev = b
            prev.wait()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            st = bs[-1].stats()
            for b in bs:
                b.free()
            return dt, st
        one_pass(3)
        dt, st = one_pass(min(steps, 8))
        out["ms_per_step"] = dt * 1e3
        out["stage_ms_last"] = {k: st[k] for k in ("ms_scan", "ms_score", "ms_group", "ms_rank", "ms_total", "ms_scan_kernel", "ms_filter_score_kernel")}
    elif mode == "rerun":
        bs = [model.encode_packed_device(dev_blob.data_ptr(), dev_blob.numel(), len(queries), params) for _ in range(2)]
        for b in bs:
            b.run(stream)
            b.run(stream)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for k in range(steps):
            b = bs[k & 1]
            if k >= 2:
                b.wait()
            b.run_async(stream)
        for b in bs:
            b.wait()
        torch.cuda.synchronize()
        out["ms_per_step"] = (time.perf_counter() - t) / steps * 1e3
        st = bs[0].stats()
        out["stage_ms_last"] = {k: st[k] for k in ("ms_scan", "ms_score", "ms_group", "ms_rank", "ms_total", "ms_scan_kernel", "ms_filter_score_kernel")}
    elif mode == "kernels":
        # every kernel alone on the GPU (ANX_RUN_OVERLAP=0, one stream): HIP-event times of the run's stages, averaged
        A.set_switch("ANX_RUN_OVERLAP", "0")
        b = model.encode_packed_device(dev_blob.data_ptr(), dev_blob.numel(), len(queries), params)
        for _ in range(10):
            b.run(stream)
        acc = {}
        t = time.perf_counter()
        for _ in range(steps):
            b.run(stream)
            st = b.stats()
            for k in ("ms_scan", "ms_score", "ms_group", "ms_rank", "ms_total", "ms_scan_kernel", "ms_filter_score_kernel"):
                acc[k] = acc.get(k, 0.0) + st[k] / steps
        out["ms_per_step"] = (time.perf_counter() - t) / steps * 1e3
        out["stage_ms"] = {k: round(v, 4) for k, v in acc.items()}
        out["tiles"] = st["n_scan_blocks"]; out["class_tests"] = st["n_class_tests"]; out["pairs"] = st["n_pairs"]; out["slots"] = st["n_pair_slots"]
        out["adj_records"] = st["n_adj_records"]; out["adj_tiles"] = st["n_adj_tiles"]
    elif mode == "small":
        out["by_batch_size"] = small_calls(A, model, queries, params)
    print(json.dumps(out))


def small_calls(A, model, queries, params, sizes=(1, 64, 1000, 32768, 1_000_000), threads=8):
    """anx_find_variants_batch (char** in, anx_result rows + offsets out) host to host: best / median microseconds per call."""
    import statistics
    import threading
    L = A.lib()
    from analiticcl_amd import _lib as LL
    cp = params._c()
    res = {}
    enc = [q.encode("utf-8") for q in queries]
    for n in sizes:
        n = min(n, len(enc))
        arr = (C.c_char_p * n)(*enc[:n])
        reps = 200 if n <= 1000 else 30 if n <= 32768 else 5
        ts = []
        for r in range(reps + 3):
            rows = C.POINTER(LL.Result)()
            offs = C.POINTER(C.c_size_t)()
            t = time.perf_counter()
            rc = L.anx_find_variants_batch(model.h, arr, n, C.byref(cp), C.byref(rows), C.byref(offs))
            dt = time.perf_counter() - t
            assert rc == 0, LL.last_error()
            L.anx_results_free(rows, offs)
            if r >= 3:
                ts.append(dt)
        res[str(n)] = {"best_us": min(ts) * 1e6, "median_us": statistics.median(ts) * 1e6, "queries_per_s_best": n / min(ts), "reps": reps}
    # 8 host threads, each issuing n = 1000 calls on the one model
    n = 1000
    per = 100
    arrs = [(C.c_char_p * n)(*enc[i * n:(i + 1) * n]) for i in range(threads)]

    def worker(a):
        for _ in range(per):
            rows = C.POINTER(LL.Result)()
            offs = C.POINTER(C.c_size_t)()
            rc = L.anx_find_variants_batch(model.h, a, n, C.byref(cp), C.byref(rows), C.byref(offs))
            assert rc == 0
            L.anx_results_free(rows, offs)
    for _pass in range(2):
        th = [threading.Thread(target=worker, args=(a,)) for a in arrs]
        t = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        dt = time.perf_counter() - t
    res["threads8_n1000"] = {"queries_per_s": threads * per * n / dt, "calls": threads * per, "s": dt}
    return res


if __name__ == "__main__":
    main()
