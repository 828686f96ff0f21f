#!/usr/bin/env python3
"""bench.py -- throughput of the variant-query hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[1]): eng.aspell lexicon (119 773 entries / 108 802 anagram classes) +
simple.alphabet, 1 M synthetic queries of length <= 16 (generator: SURVEY.md section 8(d), seed 20240601 + rank),
CLI defaults with max-edit-distance 2 (k=3, d=2, n=10, score threshold 0.25, cutoff 2.0).
A "step" = one pass of the device pipeline (anagram scan -> pair grouping -> DL/LCS/prefix/suffix scoring ->
ranking) over the whole resident query batch.  Inputs are encoded and uploaded once, before the timed region; two resident
copies alternate on one stream so that the next step is enqueued while the current one runs (no host round trip inside a run).
With N>1 every rank processes its own 1 M-query shard (weak scaling, no data-path collective) and the ranked
compact top-k records (offsets + the rows in use) are gathered to rank 0 over RCCL once per step.

Other launch forms of the same measurement:
  --single-process          ONE process drives --gpus N replicas of the lexicon through the C ABI's own split
                            (anx_model_to_devices: one host thread + stream per device, rows concatenated in input order; no
                            torch.distributed, no collective -- what a Rust / C caller of libanx gets); --replicas-on-one-gpu puts
                            all N replicas on device 0 (one-GPU boxes).
  --ranks-on-one-gpu N      dry run of the N-rank job on ONE GPU: N processes under torch.distributed.run, every rank on
                            device 0, --backend gloo (records staged through pinned host memory) or nccl (if RCCL accepts
                            ranks that share a device); with --check-gather rank 0 compares every rank's gathered export with
                            that rank's own fetch().
After the timed region (N = 1): "configs" -- the literal metric configuration (nld.aspell, ~200 k entries, len <= 16, d = 2) and
BASELINE.json configs[2], [3] (one GPU's share), [4] (one GPU's share), each behind a spot check against the oracle (--no-extras
skips them).

Prints ONE JSON line (rank 0).  value = scored (query,candidate) pairs per second, whole job.
"""
import argparse
import json
import os
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def baseline_metric() -> str:
    """BASELINE.json's metric string, verbatim."""
    try:
        with open(os.path.join(REPO, "BASELINE.json"), encoding="utf-8") as f:
            return json.load(f)["metric"]
    except Exception:
        return "scored (query,candidate) pairs/sec + queries/sec, 200k-lexicon, len\u226416, 1/2/4/8 GPU"


def usable_cores() -> int:
    """Host cores this process may actually use: the cgroup CPU quota when there is one (the GPU boxes expose 256
    hardware threads but grant 16 CPUs; more OpenMP threads than that only thrash), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()[:2]
            if quota != "max":
                n = min(n, max(1, -(-int(quota) // int(period))))
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                quota, period = int(f.read()), int(g.read())
                if quota > 0:
                    n = min(n, max(1, -(-quota // period)))
        except Exception:
            pass
    return n


def roofline_of(args, model, queries, st, scan_ms, fs_ms, total_ms):
    """The roofline object of the JSON line: the slowest kernel of the run against the HBM roof, both kernels with their
    algorithmic bytes and VALU-issue floors."""
    # ---- roofline of the dominant kernel (per launch, rank 0) --------------------------------------
    # The dominant kernel = the slowest kernel of THIS run (k_scan_bits or k_filter_score, HIP events around each launch on
    # the launch stream: anx_batch_stats.ms_scan_kernel / ms_filter_score_kernel); "per_kernel" carries both.
    # Algorithmic bytes per launch:
    #  k_scan_bits: query planes 16 B/query + tile descriptors 44 B/tile + class record, signature 44 B/class
    #               (each once per launch) + pair list out 8 B/slot;
    #  k_filter_score: what the kernel has to touch per materialised pair-list slot: pair record 8 B + query / entry
    #               symbols and lengths 16 + 8 B, plus 16 B of survivor record per pair that passes the score threshold.
    #               SURVEY.md section 8(d)'s literal figure (Lpad + 32 B for EVERY scored pair, i.e. 16 B of result per pair
    #               although only survivors are written) is reported next to it as "survey_model".
    # SURVEY.md section 8(d)'s whole-path figure, pairs*(Lpad+32) + queries*208, is reported as "pipeline".
    lpad = 16 if args.max_len <= 16 else (24 if args.max_len <= 24 else 32)
    n_classes = model.num_classes()
    # round 3: the scan's fused expansion applies the band-match bound itself, so per pair it tests it touches what the scoring
    # kernel's prefilter touched before (candidate symbols 16 B + entry meta 8 B = the `Lpad + 8` of SURVEY.md section 8(d)'s
    # per-pair figure; the query's symbols come from LDS, staged once per tile: 16 B per query) and writes a pair record only
    # for the survivors
    # round 5: the scan STREAMS the adjacency list of every tile's signature (analiticcl_amd/csrc/adjacency.h: 12 B per record, padding
    # included) instead of gathering 16-B records out of an L2-resident image: the lists are real HBM traffic.  Two figures:
    #   access bytes     = what the kernel requests: every tile's list (n_adj_records x 12), the 32-B candidate record of every pair the
    #                      fused band filter tests, queries, tile descriptors, pair-list slots out;
    #   compulsory bytes = what a launch has to move at least once: the list of every DISTINCT (length, signature) group of the batch
    #                      (n_adj_records_first: the tiles a large group is cut into re-read it, as a rule from the Infinity Cache), the
    #                      candidate records once (the e_rec image, 32 B per lexicon entry), queries, tiles, slots out.
    # `achieved` / `frac` of the JSON line are the compulsory bytes over the kernel's duration (SURVEY.md section 8(d)'s sense);
    # `access_frac` is the requested-bytes figure the line carried until round 4.
    n_entries = model.num_instances()
    fused_pairs = st.get("n_prefiltered_in_scan", 0)
    adj_rec, adj_first = st.get("n_adj_records", 0), st.get("n_adj_records_first", 0)
    scan_fixed = st["n_queries"] * (16 + 16) + st["n_scan_blocks"] * 60 + st["n_pair_slots"] * 8
    if adj_rec:
        scan_bytes = scan_fixed + adj_rec * 12 + fused_pairs * 32
        scan_compulsory = scan_fixed + adj_first * 12 + min(fused_pairs, n_entries) * 32
    else:  # ANX_SCAN_ADJ=0 / alphabets beyond the bit-plane scan: the round-4 model (16-B record gathers out of the lexicon image)
        scan_bytes = scan_fixed + n_classes * 44 + fused_pairs * (lpad + 8)
        scan_compulsory = scan_fixed + n_entries * (16 + 32)
    fs_bytes = st["n_pair_slots"] * (8 + lpad + 8) + st["n_survivors"] * 16
    fs_compulsory = st["n_pair_slots"] * 8 + (st["n_queries"] + n_entries) * 32 + st["n_survivors"] * 16
    fs_bytes_survey = min(st["n_pairs"], st["n_pair_slots"]) * (lpad + 32)
    if fs_ms > scan_ms:
        kname, kbytes, kcomp, kms = "k_filter_score", fs_bytes, fs_compulsory, fs_ms
    else:
        kname, kbytes, kcomp, kms = ("k_scan_adj" if adj_rec else "k_scan_bits"), scan_bytes, scan_compulsory, scan_ms
    achieved = kcomp / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
    access_gbs = kbytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
    pipeline_bytes = st["n_pairs"] * (lpad + 32) + st["n_queries"] * 208
    pipeline_gbs = pipeline_bytes / (total_ms * 1e-3) / 1e9 if total_ms > 0 else 0.0
    # The bound that actually binds both kernels is VALU issue (89 % / 77 % VALU-active, profiles/): the floor below counts
    # only the instructions the algorithm cannot do without, at the measured issue cost per wave-instruction per SIMD
    # (tools/ubench_valu.hip: 2-operand ops 2.2 cycles, VOP3 ops such as v_bcnt / v_alignbit / v_sad_u8 4.3), over the
    # 1024 SIMDs at 2.4 GHz.
    #  scan: per 256 class tests of T planes T*4 v_and_b32 + T*4 v_bcnt_u32_b32 + 4 v_alignbit_b32.
    SIMD_HZ = 1024 * 2.4e9
    kinds = st["n_tests_kind"]
    issue_cycles = (kinds[0] * (8 * 4.3 + 4.3) * 4 + sum(kinds[t] * (t * 4 * 6.5 + 4 * 4.3) for t in range(1, 5))) / 256.0
    valu_floor_ms = issue_cycles / SIMD_HZ * 1e3
    #  filter_score, per wave of 64 (DESIGN.md section 5 K2+K3): SWAR band filter of every slot over as many 4-symbol words as
    #  the pair needs (the kernel picks 2 / 3 / 4 per wave; estimated here from the query lengths as ceil((len + 1) / 4)):
    #  per word the unshifted comparison + 2d shifted ones + 2 x (and, bcnt) popcounts.  Alphabets of <= 124 classes (7-bit
    #  symbol codes, the kernel's B7 instances): 2 mask ands, unshifted (xor, add, and, and) = 8.8 cycles, shifted
    #  (alignbyte, xor, add, and | alignbyte, and) = 17.4; otherwise unshifted (xor, and, add, or, and, and) = 15.4, shifted
    #  (alignbyte, xor, and, add, or3, and | alignbyte, and) = 24.0.
    #  DL of every selected pair, round 6: by diagonals on mismatch masks (kernels_score.hpp dl_diag): 2d + 1 masks from the symbol
    #  planes (6 x (shift, xor-or) + 1 at 2.3 cycles = 30) + (d + 1)^2 furthest-reaching cells (slide = shift 2.3 + ffbl 4.2 + add 2.3,
    #  max3 4.3, two adds 4.6 = 17.7) + one transposition test per (e, a, b, k') (bfe 4.1 + add 2.3 = 6.4; 1 / 6 / 20 of them for
    #  d = 1 / 2 / 3) + the final select (2 x 4.2 per cell) -- no row loop: the same for every length (until round 6: rows x (2d+1)
    #  cells x 16 cycles, ~720 a wave at d = 2);
    #  tail of every DL survivor: the masks again (30 x (2d + 1)), prefix / suffix (ffbl, ffbh + shifts: 20), the band's runs
    #  (mean query length x 3 registers x 2 x 2.3), the f64 score (5 quotients from LDS, 9 f64 operations at 4.3, compare: 80) = ~350 cycles
    #  at d = 2 (until round 6: byte loops through LDS, ~900).
    dd = args.edit_distance
    nw = 4 if args.max_len <= 16 else 8
    sample_q = queries[:20000]
    mean_len = sum(len(q) for q in sample_q) / max(len(sample_q), 1)
    words = sum(min(nw, (len(q) + 1 + 3) // 4) for q in sample_q) / max(len(sample_q), 1)
    from analiticcl_amd import _lib as _L
    b7 = _L.lib().anx_model_alphabet_size(model.h) < 0x7E  # = classes + 1 = the largest symbol code (unknown): engine.hip's condition
    c_mask, c_unshifted, c_shifted = (4.4, 8.8, 17.4) if b7 else (0.0, 15.4, 24.0)
    band_cycles_per_wave = (c_mask + c_unshifted + 2 * dd * c_shifted + 2 * 6.5) * words
    fused = st.get("n_prefiltered_in_scan", 0)
    # round 3: with the filter fused into the scan's expansion the band bound of those pairs is the SCAN's work (uniform d per
    # tile: one OR per shifted word less); k_filter_score only filters the few pairs the scan left unflagged (wide candidates)
    fs_filter_waves = 0.0 if fused else st["n_pair_slots"] / 64.0
    ntrans = {0: 0, 1: 1, 2: 6, 3: 20}.get(dd, 20)
    dl_cycles = (2 * dd + 1) * 30.0 + (dd + 1) ** 2 * (17.7 + 8.4) + ntrans * 6.4
    tail_cycles = (2 * dd + 1) * 30.0 + 20.0 + mean_len * 3 * 2 * 2.3 + 80.0
    fs_cycles = fs_filter_waves * band_cycles_per_wave + (st["n_selected"] / 64.0) * dl_cycles + (st["n_survivors"] / 64.0) * tail_cycles
    fs_valu_floor_ms = fs_cycles / SIMD_HZ * 1e3
    valu_floor_ms += (fused / 64.0) * ((c_mask + c_unshifted + 2 * dd * (c_shifted - 2.2) + 2 * 6.5) * words) / SIMD_HZ * 1e3
    # HBM bytes per launch of that kernel from the committed PMC passes: only for the same workload AND the same kernel
    # sources (the profile is tagged with a hash of csrc/*.hip, *.hpp; stale numbers are dropped)
    traffic, traffic_src = None, None
    try:
        if (args.lexicon, args.max_len, args.anagram_distance, args.edit_distance, args.queries) == ("eng", 16, 3, 2, 1_000_000):
            import glob
            import hashlib
            h = hashlib.sha256()
            for f in sorted(glob.glob(os.path.join(REPO, "analiticcl_amd", "csrc", "*.hip")) + glob.glob(os.path.join(REPO, "analiticcl_amd", "csrc", "*.hpp"))):
                with open(f, "rb") as fh:
                    h.update(fh.read())
            for cand in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic.json")), reverse=True):
                with open(cand) as f:
                    pj = json.load(f)
                if pj.get("kernel_src_sha256") == h.hexdigest():
                    traffic, traffic_src = pj["kernels"][kname]["traffic_bytes"], os.path.basename(cand)
                    break
        elif getattr(args, "profile_key", None):  # the other configurations: profiles/r*_pmc_traffic.json "configs" (tools/collect_profiles.py)
            import glob
            for cand in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic.json")), reverse=True):
                with open(cand) as f:
                    pj = json.load(f)
                ent = pj.get("configs", {}).get(args.profile_key, {}).get(kname)
                if ent:
                    traffic, traffic_src = ent["traffic_bytes"], os.path.basename(cand) + ":" + args.profile_key
                    break
    except Exception:
        traffic = None
    roofline = {"bound": "valu_issue", "hbm_bound_fields": "achieved / peak / unit / frac = compulsory HBM bytes per launch of the slowest kernel over its duration "
                                                            "(the figure the bench contract asks for); what binds the kernel is vector-instruction issue",
                "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "avg_kernel_ms": kms,
                "compulsory_bytes": kcomp, "access_bytes": kbytes, "access_gbs": access_gbs, "access_frac": access_gbs / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": kcomp,
                "note": "integer scan / DL path: both kernels are bound by vector-instruction issue (DESIGN.md section 5); valu_issue_frac = algorithmic "
                        "instruction floor / measured kernel time.  The scan streams its adjacency lists from HBM (12 B per record): access_bytes "
                        "counts every tile's list, compulsory_bytes the list of every distinct (length, signature) group once",
                "kernels_ms": {"k_scan_bits": scan_ms, "k_filter_score": fs_ms},
                "per_kernel": {name: {"avg_kernel_ms": ms, "compulsory_bytes": comp, "access_bytes": nbytes, "algorithmic_bytes_per_launch": comp,
                                      "achieved": (comp / (ms * 1e-3) / 1e9 if ms > 0 else 0.0),
                                      "frac": (comp / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else 0.0),
                                      "access_frac": (nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else 0.0),
                                      "bound": "valu_issue", "valu_issue_floor_ms": fl, "valu_issue_frac": (fl / ms if ms > 0 else 0.0)}
                               for name, nbytes, comp, ms, fl in ((("k_scan_adj (+ k_scan_bits)" if adj_rec else "k_scan_bits"), scan_bytes, scan_compulsory, scan_ms, valu_floor_ms),
                                                                  ("k_filter_score", fs_bytes, fs_compulsory, fs_ms, fs_valu_floor_ms))},
                "k_filter_score_survey_model": {"algorithmic_bytes_per_launch": fs_bytes_survey,
                                                "frac": (fs_bytes_survey / (fs_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if fs_ms > 0 else 0.0)},
                "pipeline_algorithmic_bytes": pipeline_bytes, "pipeline_gbs": pipeline_gbs,
                "pipeline_frac": pipeline_gbs / HBM_PEAK_GBS,
                "scan_valu_issue_floor_ms": valu_floor_ms,
                "scan_valu_issue_frac": valu_floor_ms / scan_ms if scan_ms > 0 else 0.0,
                "scan_class_tests_per_s": st["n_class_tests"] / (scan_ms * 1e-3) if scan_ms > 0 else 0.0,
                "scan_tests_by_planes": kinds, "scan_tiles": st["n_scan_blocks"]}
    return roofline


def e2e_of(args, model, queries, params, stream_handle, torch):
    """End to end from ONE host buffer of NUL-terminated strings to the ranked rows in host memory (never `value`)."""
    e2e = None
    if True:
        reps = []
        # the queries as ONE host buffer, every string followed by a NUL byte: what a caller that reads its input from a
        # file or a socket holds (the reference's CLI reads lines the same way, src/bin/analiticcl.rs:416-448); building it
        # from a Python list of str costs more than the whole pipeline and is not part of the boundary
        packed = ("\0".join(queries) + "\0").encode("utf-8")
        for _ in range(3):
            t = time.perf_counter()
            b2 = model.encode_packed(packed, len(queries), params)
            t1 = time.perf_counter()
            b2.run(stream_handle)
            t2 = time.perf_counter()
            arrs = b2.fetch_compact()   # 16-byte records + u32 offsets (anx_batch_fetch_compact)
            t3 = time.perf_counter()
            b2.free()
            reps.append((t3 - t, t1 - t, t2 - t1, t3 - t2, int(arrs[0][-1])))
            del arrs  # the rows live in a pinned buffer of the library's cache: released here, reused by the next fetch
        best = min(reps)
        # the same with two host threads, each running encode -> run -> fetch on its own batches and its own stream (the
        # library is thread-safe on one model): uploads, kernels and downloads of different batches overlap
        import threading
        nthr, per = 2, 8
        streams2 = [torch.cuda.Stream() for _ in range(nthr)]
        def worker(st2):
            for _ in range(per):
                bb = model.encode_packed(packed, len(queries), params)
                bb.run(st2.cuda_stream)
                res = bb.fetch_compact()
                del res
                bb.free()
        th = [threading.Thread(target=worker, args=(x,)) for x in streams2]  # warm the pools of a second set of buffers
        for x in th:
            x.start()
        for x in th:
            x.join()
        t = time.perf_counter()
        th = [threading.Thread(target=worker, args=(x,)) for x in streams2]
        for x in th:
            x.start()
        for x in th:
            x.join()
        piped = nthr * per * args.queries / (time.perf_counter() - t)
        # ONE caller thread, anx_pipeline (three library threads: encode(i + 2) / run(i + 1) / fetch(i) in flight on separate
        # streams); every returned batch is compared with the synchronous path's rows (offsets, ids, scores)
        import hashlib

        import analiticcl_amd as A_

        def digest(arrs):
            h = hashlib.sha256()
            h.update(arrs[0].tobytes())
            h.update(arrs[1].tobytes())
            return h.hexdigest()
        bref = model.encode_packed(packed, len(queries), params)
        bref.run(stream_handle)
        want = digest(bref.fetch_compact())
        bref.free()
        PD = 6  # jobs in flight (2-6 measured the same; 8 outgrows the pinned result cache of 1 GB and halves the rate)
        pl = A_.Pipeline(model, depth=PD)
        got_ok = True

        def pipe_pass(njobs, check):
            nonlocal got_ok
            t = time.perf_counter()
            sub, last = 0, None
            for _k in range(njobs):
                pl.submit(packed, len(queries), params)
                sub += 1
                if sub >= PD:
                    last = pl.next()
                    sub -= 1
                    if check:
                        got_ok = digest(last) == want and got_ok
            while sub:
                last = pl.next()
                sub -= 1
                if check:
                    got_ok = digest(last) == want and got_ok
            dt = time.perf_counter() - t
            got_ok = digest(last) == want and got_ok
            pipe_last[0] = (last[0].copy(), last[1].copy())
            return njobs * args.queries / dt
        pipe_last = [None]
        pipe_pass(8, True)   # every batch of this pass is checked (and the pools of the extra buffers warm up); hashing 70 MB per
        passes = sorted(pipe_pass(40, False) for _ in range(5))  # best and median of five passes of 40 jobs (filling and draining the
        pipelined, pipelined_median = passes[-1], passes[2]     # pipeline is inside the timed pass: ~6 % at this length)
        # batch would dominate a timed loop: there the last batch stands for all
        pl.close()
        e2e = {"queries_per_s": args.queries / best[0], "two_threads_queries_per_s": piped,
               "pipelined_queries_per_s": pipelined, "pipelined_median_queries_per_s": pipelined_median, "pipelined_parity": "ok (every batch's rows equal the synchronous path's)" if got_ok else "MISMATCH", "s_per_batch": best[0], "encode_upload_s": best[1], "run_s": best[2],
               "download_s": best[3], "rows": best[4], "pipelined_last": pipe_last[0],
               "what": "host buffer of NUL-terminated UTF-8 strings -> anx_batch_encode_packed (H2D + device-side encoder) -> anx_batch_run -> "
                       "anx_batch_fetch_compact (ranked rows in input order as 16-byte records + u32 offsets, pinned host memory), best of 3, one batch at a time, no overlap between batches"}
    return e2e


def by_batch_size_of(model, om, queries, params, op, sizes=(1, 64, 1000, 32768, 1_000_000), threads=8, alphabet_path=None, lexicon_path=None):
    """The call at the reference's own granularity (find_variants takes ONE string, src/lib.rs:972; the CLI and the Python binding fan out
    in batches of 1 000, src/bin/analiticcl.rs:416, bindings/python/src/lib.rs:704-749): anx_find_variants_batch (char** in, anx_result
    rows + offsets out) host to host for n = 1 .. 1 M inputs -- best / median microseconds per call, the rows of every size against the
    oracle -- and `threads` host threads each issuing calls of 1 000 inputs on the one model.  Calls of <= 4096 short inputs take the
    small path (analiticcl_amd/csrc/small_path.hpp: nine launches, one host wait); larger ones the batch pipeline."""
    import ctypes as C
    import random
    import statistics
    import threading

    import analiticcl_amd as A
    from analiticcl_amd import _lib as LL
    L = A.lib()
    cp = params._c()
    enc = [q.encode("utf-8") for q in queries]
    res = {}

    def small_taken():
        out = (C.c_uint64 * 2)()
        L.anx_debug_small_stats(out)
        return out[0]
    for n in sizes:
        n = min(n, len(enc))
        arr = (C.c_char_p * n)(*enc[:n])
        reps = 200 if n <= 1000 else 30 if n <= 32768 else 5
        ts = []
        t_small = small_taken()
        for r in range(reps + 3):
            rows = C.POINTER(LL.Result)()
            offs = C.POINTER(C.c_size_t)()
            t = time.perf_counter()
            rc = L.anx_find_variants_batch(model.h, arr, n, C.byref(cp), C.byref(rows), C.byref(offs))
            dt = time.perf_counter() - t
            if rc != 0:
                raise RuntimeError(LL.last_error())
            if r == reps + 2:   # the last call's rows against the oracle (a sample of the larger sizes)
                idx = list(range(n)) if n <= 64 else random.Random(n).sample(range(n), 200)
                for i in idx:
                    got = [(rows[j].vocab_id, rows[j].dist_score, rows[j].freq_score) for j in range(offs[i], offs[i + 1])]
                    if got != om.find_variants(queries[i], op):
                        raise RuntimeError(f"by_batch_size n={n}: rows of {queries[i]!r} differ from the oracle's")
            L.anx_results_free(rows, offs)
            if r >= 3:
                ts.append(dt)
        res[str(n)] = {"best_us": min(ts) * 1e6, "median_us": statistics.median(ts) * 1e6, "queries_per_s_best": n / min(ts), "calls": reps,
                       "path": "small" if small_taken() - t_small == reps + 3 else "batch", "parity": f"ok ({min(n, 200) if n > 64 else n} queries vs the oracle)"}
    n, per = min(1000, len(enc) // threads), 100
    arrs = [(C.c_char_p * n)(*enc[i * n:(i + 1) * n]) for i in range(threads)]

    def worker(a):
        for _ in range(per):
            rows = C.POINTER(LL.Result)()
            offs = C.POINTER(C.c_size_t)()
            if L.anx_find_variants_batch(model.h, a, n, C.byref(cp), C.byref(rows), C.byref(offs)) != 0:
                raise RuntimeError("threads: call failed")
            L.anx_results_free(rows, offs)
    best = 0.0
    for _pass in range(3):
        th = [threading.Thread(target=worker, args=(a,)) for a in arrs]
        t = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        best = max(best, threads * per * n / (time.perf_counter() - t))
    res[f"threads{threads}_n{n}"] = {"queries_per_s": best, "calls_per_pass": threads * per,
                                     "what": f"{threads} PYTHON host threads, each {per} calls of {n} inputs on the one model (ctypes releases the GIL for the call, the "
                                             "interpreter's own work per call is serialised by it), best of 3 passes"}
    # the same with NATIVE host threads: tools/small_threads.cpp against libanx.so, as a child process with a model of its own
    if alphabet_path is None or lexicon_path is None:
        return res
    try:
        import subprocess
        import tempfile as _tf
        d_ = _tf.mkdtemp(prefix="anx_small_threads_")
        exe = os.path.join(d_, "small_threads")
        libdir = os.path.join(REPO, "analiticcl_amd")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(REPO, "include"), os.path.join(REPO, "tools", "small_threads.cpp"), "-o", exe,
                               "-L", libdir, "-lanx", f"-Wl,-rpath,{libdir}"], stderr=subprocess.DEVNULL)
        qf = os.path.join(d_, "queries.txt")
        with open(qf, "w", encoding="utf-8") as f:
            f.write("\n".join(queries[:threads * n]) + "\n")
        r_ = subprocess.run([exe, alphabet_path, lexicon_path, qf, str(threads), str(n), "200"], capture_output=True, text=True, timeout=300)
        if r_.returncode == 0:
            res[f"native_threads{threads}_n{n}"] = json.loads(r_.stdout.strip().splitlines()[-1])
        else:
            res[f"native_threads{threads}_n{n}"] = {"error": (r_.stderr or r_.stdout)[-300:]}
    except Exception as e_:  # noqa: BLE001
        res[f"native_threads{threads}_n{n}"] = {"error": repr(e_)[:300]}
    return res


def cpu_baseline_of(args, paths, queries, ncores, seconds=15.0):
    """The C oracle ("port" of the reference algorithm, oracle/anx_oracle.c) on this box's host cores: a bounded sample of the
    same queries.  Test infrastructure used as the reported baseline only."""
    cpu = None
    if True:
        from oracle import cwrap as O
        om = O.OracleModel(alphabet_path=paths["alphabet"])
        om.read_lexicon(paths[args.lexicon])
        om.build()
        op = O.make_params(("abs", args.anagram_distance), ("abs", args.edit_distance), 10, 0.25, 2.0)
        if args.cpu_sample > 0:
            sample = min(args.cpu_sample, args.queries)
        else:  # calibrate on a short run, then size the sample for ~15 s of wall time
            ncal = min(args.queries, 16 * ncores)
            t = time.perf_counter()
            om.find_variants_batch(queries[:ncal], op, nthreads=ncores, stride=16)
            rate = ncal / max(time.perf_counter() - t, 1e-3)
            sample = int(max(ncal, min(args.queries, rate * seconds)))
        t = time.perf_counter()
        rc, _res, _counts, cpairs, _ccls = om.find_variants_batch(queries[:sample], op, nthreads=ncores, stride=16)
        dt = time.perf_counter() - t
        # one thread as well: a short prefix of the same sample, ~5 s
        n1 = int(max(64, min(sample, (sample / dt) / ncores * 5.0)))
        t = time.perf_counter()
        _rc, _r, _c, cpairs1, _cc = om.find_variants_batch(queries[:n1], op, nthreads=1, stride=16)
        dt1 = time.perf_counter() - t
        cpu = {"value": cpairs / dt, "unit": "pairs/s", "cores": ncores, "kind": "port",
               "single_thread": {"value": cpairs1 / dt1, "queries_per_s": n1 / dt1, "sample": f"first {n1} queries, {dt1:.1f} s"},
               "queries_per_s": sample / dt,
               "sample": f"first {sample} of the same {args.queries} queries, C oracle (oracle/anx_oracle.c), "
                         f"OpenMP dynamic schedule, {ncores} threads (= usable cores: cgroup quota of {os.cpu_count()} hardware threads), {dt:.1f} s"}
    return cpu



# ---- per-config numbers measured after the timed region (never `value`) ---------------------------------------------------------
def _spot_check(model, om, queries, arrays, op, n, rescore=None):
    """n sampled queries of a finished batch against the oracle (test infrastructure used as the checker): ranked ids and f64
    scores.  Returns 'ok (n queries)' or raises."""
    import random
    off, vid, dist, freq = arrays
    idx = random.Random(5).sample(range(len(queries)), n)
    for i in idx:
        exp = om.find_variants(queries[i], op)
        if rescore:
            exp = rescore(exp, queries[i])
        got = [(int(vid[j]), float(dist[j]), float(freq[j])) for j in range(off[i], off[i + 1])]
        if [v for v, _d, _f in got] != [v for v, _d, _f in exp] or any(abs(a[1] - b[1]) > 1e-6 or abs(a[2] - b[2]) > 1e-6 for a, b in zip(got, exp)):
            raise RuntimeError(f"parity spot check failed on {queries[i]!r}: {got[:3]} vs {exp[:3]}")
    return f"ok ({n} queries vs the oracle)"


def _profile_traffic(profile_key, kernel):
    """HBM bytes per launch of `kernel` in the tracked PMC passes of a configuration (profiles/r*_pmc_traffic.json "configs",
    written by tools/collect_profiles.py from profiles/r*_{conf,big,search}_pmc.md) -> (bytes | None, source | None)."""
    import glob
    short = kernel.split(" ")[0]
    for cand in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            with open(cand) as f:
                ent = json.load(f).get("configs", {}).get(profile_key, {}).get(short)
        except Exception:
            ent = None
        if ent:
            return ent["traffic_bytes"], os.path.basename(cand) + ":" + profile_key
    return None, None


def _config_roofline(model, queries, st, total_ms, lexicon, max_len, dd, nq, extra=None, profile_key=None):
    """The roofline entry of an extra configuration: the slowest kernel of its device pass against the 8 TB/s HBM roof, from the
    same byte / instruction models as the headline line (roofline_of) and the live HIP-event kernel times of anx_batch_stats;
    `extra` = (name, ms, algorithmic bytes, note) of a kernel outside that pair (k_conf_script, k_lattice: timed by the library's
    kernel timer, anx_debug_kernel_time)."""
    r = roofline_of(argparse.Namespace(max_len=max_len, edit_distance=dd, anagram_distance=3, lexicon=lexicon, queries=nq), model, queries, st,
                    st["ms_scan_kernel"], st["ms_filter_score_kernel"], total_ms)
    pk = r["per_kernel"]
    # (name, ms, compulsory bytes, access bytes, VALU-issue fraction, what binds it, note)
    cands = [(k, v["avg_kernel_ms"], v["compulsory_bytes"], v["access_bytes"], v["valu_issue_frac"], "valu_issue", None) for k, v in pk.items()]
    if extra:  # (name, ms, bytes, note[, bound])
        cands.append((extra[0], extra[1], extra[2], extra[2], None, extra[4] if len(extra) > 4 else "divergence", extra[3]))
    k, ms, comp, acc, vf, bound, note = max(cands, key=lambda c: c[1])
    traffic, tsrc = _profile_traffic(profile_key, k) if profile_key else (None, None)
    gbs = lambda nb: (nb / (ms * 1e-3) / 1e9 if ms > 0 else 0.0)  # noqa: E731
    out = {"bound": bound, "kernel": k, "avg_kernel_ms": ms, "compulsory_bytes": comp, "access_bytes": acc, "algorithmic_bytes": comp,
           "achieved": gbs(comp), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs(comp) / HBM_PEAK_GBS, "access_frac": gbs(acc) / HBM_PEAK_GBS,
           "valu_issue_frac": vf, "kernels_ms": {c[0]: c[1] for c in cands}, "traffic": traffic, "traffic_source": tsrc,
           "hbm_bound_fields": "achieved / frac = compulsory HBM bytes of the slowest kernel over its duration; `bound` names what binds it"}
    if note:
        out["note"] = note
    return out


def _time_runs(b, reps=5):
    b.run()
    b.run()
    t = time.perf_counter()
    for _ in range(reps):
        b.run()
    return (time.perf_counter() - t) / reps


def _e2e_packed(model, queries, params, reps=3):
    packed = ("\0".join(queries) + "\0").encode("utf-8")
    best = None
    for _ in range(reps):
        t = time.perf_counter()
        b = model.encode_packed(packed, len(queries), params)
        b.run()
        arrs = b.fetch_arrays()
        dt = time.perf_counter() - t
        b.free()
        del arrs
        best = dt if best is None else min(best, dt)
    return len(queries) / best


def _cpu_baseline_config(om, qs, op, ncores, st, seconds=8.0, post=None, post_what=""):
    """The C oracle (a "port" of the reference algorithm; oracle/anx_oracle.c, OpenMP batch entry) over a bounded prefix of an extra
    configuration's queries on this box's host cores.  post(rows of query i, query) -> rows: what the configuration does on top
    (configs[2]: the confusable weighting of oracle/confusable_oracle.py, Python), timed with it.  st: the GPU run's statistics (pairs per query)."""
    from oracle import cwrap as O
    ncal = min(len(qs), 16 * ncores)
    t = time.perf_counter()
    O.batch_rows(om, qs[:ncal], op, nthreads=ncores, stride=32)
    rate = ncal / max(time.perf_counter() - t, 1e-3)
    n = int(max(ncal, min(len(qs), rate * seconds)))
    t = time.perf_counter()
    c, ov, od, of, tp, _tc = O.batch_rows(om, qs[:n], op, nthreads=ncores, stride=32)
    dt_c = time.perf_counter() - t
    dt_post = 0.0
    if post is not None:
        t = time.perf_counter()
        for i in range(n):
            post([(int(ov[i, j]), float(od[i, j]), float(of[i, j])) for j in range(c[i])], qs[i])
        dt_post = time.perf_counter() - t
    dt = dt_c + dt_post
    return {"value": tp / dt, "unit": "pairs/s", "queries_per_s": n / dt, "cores": ncores, "kind": "port",
            "sample": f"first {n} of the configuration's {len(qs)} queries, C oracle (oracle/anx_oracle.c, OpenMP, {ncores} threads) {dt_c:.1f} s"
                      + (f" + {post_what} {dt_post:.1f} s (one Python thread)" if post is not None else "")}


def extra_configs(args, paths, device, ncores):
    """BASELINE.json's other configurations and the literal metric configuration on this one GPU, each behind a parity spot check.
    Every entry carries its own wall time ("took_s"); an entry that fails says why instead of stopping the bench."""
    import analiticcl_amd as A
    from analiticcl_amd import synth
    from analiticcl_amd import _lib as L_
    from oracle import confusable_oracle as CO
    from oracle import cwrap as O
    out = {}

    def guarded(name, fn):
        t = time.perf_counter()
        try:
            r = fn()
        except Exception as e:  # noqa: BLE001
            r = {"error": repr(e)[:300]}
        r["took_s"] = round(time.perf_counter() - t, 1)
        out[name] = r

    nld_words = synth.load_lexicon_words(paths["nld"])
    std = dict(max_matches=10, score_threshold=0.25, cutoff_threshold=2.0)
    nspot = max(1, args.spot_check)

    def nld_len16_d2():  # the metric string's "200k-lexicon, len<=16": nld.aspell has 222 908 entries
        m = A.VariantModel(paths["alphabet"], A.Weights(), device=device)
        m.read_lexicon(paths["nld"])
        m.build()
        qs = synth.make_queries(nld_words, 1_000_000, max_len=16, seed=synth.SEED + 1)
        p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, **std)
        b = m.encode_batch(qs, p)
        dt = _time_runs(b)
        st = b.stats()
        om = O.OracleModel(alphabet_path=paths["alphabet"])
        om.read_lexicon(paths["nld"])
        om.build()
        chk = _spot_check(m, om, qs, b.fetch_arrays(), O.make_params(("abs", 3), ("abs", 2), 10, 0.25, 2.0), nspot)
        b.free()
        cpu1 = _cpu_baseline_config(om, qs, O.make_params(("abs", 3), ("abs", 2), 10, 0.25, 2.0), ncores, st, seconds=6.0)
        return {"workload": "nld.aspell (222 908 entries) + simple.alphabet, 1 M queries len<=16, k=3 d=2 n=10", "ms_per_step": dt * 1e3, "cpu_baseline": cpu1,
                "pairs_per_s": st["n_pairs"] / dt, "queries_per_s": st["n_queries"] / dt, "pairs_per_query": st["n_pairs"] / max(st["n_queries"], 1),
                "scan_kernel_ms": st["ms_scan_kernel"], "filter_score_kernel_ms": st["ms_filter_score_kernel"], "parity": chk,
                "roofline": _config_roofline(m, qs, st, dt * 1e3, "nld", 16, 2, 1_000_000)}

    def configs2():  # nld, len <= 24, d = 3, confusable weighting
        conf = os.path.join(synth.GOLDEN_DATA, "confusables10.tsv")
        m = A.VariantModel(paths["alphabet"], A.Weights(), device=device)
        m.read_lexicon(paths["nld"])
        m.read_confusablelist(conf)
        m.build()
        qs = synth.make_queries(nld_words, 1_000_000, max_len=24, seed=synth.SEED + 2)
        p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=3, **std)
        b = m.encode_batch(qs, p)
        dt = _time_runs(b)
        L_.kernel_timer(True)
        b.run()
        conf_ms, conf_n = L_.kernel_time("k_conf_script")
        L_.kernel_timer(False)
        st = b.stats()
        # k_conf_script, algorithmic bytes per scripted row: the input's bytes (<= 24) + the candidate's code points (4 B each) + its
        # character-set record (16 B) + the ranked row read (24 B) + the weight written (8 B)
        mean_len = sum(len(q) for q in qs[:20000]) / 20000.0
        conf_bytes = st["n_conf_scripts"] * (mean_len + 4 * mean_len + 16 + 24 + 8)
        om = O.OracleModel(alphabet_path=paths["alphabet"])
        om.read_lexicon(paths["nld"])
        om.build()
        confs = CO.read_confusables(conf)
        chk = _spot_check(m, om, qs, b.fetch_arrays(), O.make_params(("abs", 3), ("abs", 3), 10, 0.25, 0.0), nspot,
                          rescore=lambda exp, q: CO.late_rescore(exp, q, confs, om.text, 0.0, 2.0))
        b.free()
        e2e = _e2e_packed(m, qs, p)
        cpu2 = _cpu_baseline_config(om, qs, O.make_params(("abs", 3), ("abs", 3), 10, 0.25, 0.0), ncores, st,
                                    post=lambda rows_, q_: CO.late_rescore(rows_, q_, confs, om.text, 0.0, 2.0), post_what="confusable weighting (oracle/confusable_oracle.py)")
        return {"workload": "BASELINE.json configs[2]: nld.aspell, 1 M queries len<=24, k=3 d=3 n=10, 10 confusable patterns", "device_ms": dt * 1e3, "cpu_baseline": cpu2,
                "pairs_per_s": st["n_pairs"] / dt, "e2e_queries_per_s": e2e, "parity": chk, "conf_scripts": st["n_conf_scripts"],
                "roofline": _config_roofline(m, qs, st, dt * 1e3, "nld", 24, 3, 1_000_000,
                                             extra=("k_conf_script", conf_ms / max(conf_n, 1), conf_bytes,
                                                    "one lane group per ranked row through a branchy edit-script algorithm: bound by lane divergence, not by bytes (profiles/)", "divergence"),
                                             profile_key="conf"),
                "what": "device_ms = one pass of the device pipeline over the resident batch; e2e = packed host buffer -> encode -> run -> fetch incl. the confusable rescoring"}

    def configs3_share():  # merged 1 M-entry lexicon, one GPU's 1.25 M of the 10 M length-bucketed queries
        words = list(dict.fromkeys(synth.load_lexicon_words(paths["eng"]) + nld_words))
        lex = synth.make_lexicon(words, 1_000_000, seed=11)
        path = os.path.join(tempfile.gettempdir(), f"anx_bench_big_{os.getuid()}.lexicon")
        with open(path, "w", encoding="utf-8") as f:
            f.write("\n".join(lex) + "\n")
        m = A.VariantModel(paths["alphabet"], A.Weights(), device=device)
        m.read_lexicon(path)
        m.build()
        qs = synth.make_queries(lex, 1_250_000, max_len=32, min_len=4, seed=5)
        qs.sort(key=len)
        p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, **std)
        b = m.encode_batch(qs, p)
        dt = _time_runs(b, reps=3)
        st = b.stats()
        om = O.OracleModel(alphabet_path=paths["alphabet"])
        om.read_lexicon(path)
        om.build()
        chk = _spot_check(m, om, qs, b.fetch_arrays(), O.make_params(("abs", 3), ("abs", 2), 10, 0.25, 2.0), nspot)
        b.free()
        import random as _random
        cpu3 = _cpu_baseline_config(om, _random.Random(3).sample(qs, 60_000), O.make_params(("abs", 3), ("abs", 2), 10, 0.25, 2.0), ncores, st)   # (a random sample: the queries are sorted by length)
        del om
        # The WHOLE 10 M-query job cut the way a multi-device model cuts it (anx_model_to_devices: length-partitioned split into 8
        # cost-balanced shares, anx_debug_length_split), every share run on this one GPU one after the other: the longest share is
        # what an 8-GPU job takes.  Round 0 = the FIRST call (the split's prior alone: records per query of every (length, signature
        # class) from the adjacency lists, fill of the scan tiles); round 1 = the next call, after the split's cost model has seen
        # the shares' device times.  Same methodology as tools/length_shares.py (which runs more rounds and prints every share).
        import numpy as np
        job = synth.make_queries(lex, 10_000_000, max_len=32, min_len=4, seed=6)
        rounds = []
        for _round in range(2):
            gid = m.length_split(job, p, 8)
            share_ms, share_n = [], []
            for g_ in range(8):
                ix = np.nonzero(gid == g_)[0]
                bs = m.encode_batch([job[i] for i in ix], p)
                share_ms.append(_time_runs(bs, reps=2) * 1e3)
                share_n.append(int(ix.size))
                bs.free()
            rounds.append({"share_ms": share_ms, "share_queries": share_n, "sum_ms": sum(share_ms), "longest_share_ms": max(share_ms),
                           "balance": sum(share_ms) / 8 / max(share_ms)})
            if _round == 0:
                m.length_split(job, p, 8, learn_ms=share_ms)
        del job
        os.unlink(path)
        return {"workload": "BASELINE.json configs[3], one GPU's share: merged 1 M-entry synthetic lexicon, 1.25 M of the 10 M length-bucketed queries len 4-32, k=3 d=2 n=10",
                "ms_per_batch": dt * 1e3, "ms_per_1M_queries": dt * 1e3 / 1.25, "pairs_per_s": st["n_pairs"] / dt, "queries_per_s": st["n_queries"] / dt, "cpu_baseline": cpu3,
                "scan_kernel_ms": st["ms_scan_kernel"], "filter_score_kernel_ms": st["ms_filter_score_kernel"], "scan_tiles": st["n_scan_blocks"], "parity": chk,
                "roofline": _config_roofline(m, qs, st, dt * 1e3, "big", 32, 2, 1_250_000, profile_key="big"),
                "what": "ms_per_1M_queries = a RANDOM eighth of the job (what consecutive input ranges give a GPU: an eighth of every (length, signature) group); "
                        "by_length = the 8 shares of the whole job under the length-partitioned split the library uses for multi-device models",
                "by_length": {"workload": "the whole 10 M-query job as the 8 shares of the length-partitioned split, run one after the other on this GPU (tools/length_shares.py's methodology)",
                              "first_call_balance": rounds[0]["balance"], "first_call_longest_share_ms": rounds[0]["longest_share_ms"],
                              "second_call_balance": rounds[1]["balance"], "second_call_longest_share_ms": rounds[1]["longest_share_ms"],
                              "whole_job_sum_ms": rounds[1]["sum_ms"], "job_speedup_vs_consecutive_ranges": dt * 1e3 / rounds[1]["longest_share_ms"],
                              "rounds": rounds}}

    def configs4_share():  # search mode: one GPU's 12.5 MB of the 100 MB running text, n-gram windows + bigram LM
        import random
        from oracle import twin as T
        if os.path.join(REPO, "tests") not in sys.path:
            sys.path.insert(0, os.path.join(REPO, "tests"))
        eng_words = synth.load_lexicon_words(paths["eng"])
        m = A.VariantModel(paths["alphabet"], A.Weights(), device=device)
        m.read_lexicon(paths["eng"])
        rng = random.Random(7)
        common = [w for w in eng_words if w.isalpha()][::23][:5000]
        lm = [(f"{rng.choice(common)} {rng.choice(common)}", rng.randrange(1, 20)) for _ in range(20000)]
        lm += [(f"<bos> {w}", 5) for w in common[:500]]
        LM = A.VocabParams(vocabtype="LM")
        for t_, f_ in lm:
            m.add_to_vocabulary(t_, f_, LM)
        m.build()
        texts = synth.make_running_text(common, 12.5, seed=7)
        nbytes = sum(len(t_.encode("utf-8")) for t_ in texts)
        sp = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10, max_ngram=3, **{k: v for k, v in std.items() if k != "max_matches"})
        m.find_all_matches_arrays(texts[:2000], sp)  # warm-up (device pool, pinned buffers)
        # the C entry point alone (what a Rust / C caller sees), then once more through the numpy view for the parity check
        import ctypes as C
        from analiticcl_amd import _lib as L
        arr = (C.c_char_p * len(texts))(*[t_.encode("utf-8") for t_ in texts])
        spc = sp._c_search()
        best = None
        call_s = []

        def _throttled():  # CFS quota: periods in which the cgroup ran out of CPU time (a search call keeps ~10 host cores busy; a
            for path_ in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):   # throttled period stalls it for tens of ms)
                try:
                    for line_ in open(path_):
                        if line_.startswith("nr_throttled"):
                            return int(line_.split()[1])
                except OSError:
                    pass
            return 0
        thr0 = 0
        # a stream of calls: the first ones of a process still grow the pinned result cache and the device pool (a same-box series:
        # best of calls 1-5 256 MB/s, of 6-10 274, of 11-15 284); four untimed calls, then five timed ones (kernel timer on)
        for i_ in range(9):
            if i_ == 4:
                L_.kernel_timer(True)
                thr0 = _throttled()
            ms, offs, rows, nrows = C.POINTER(L.Match)(), C.POINTER(C.c_size_t)(), C.POINTER(L.Result)(), C.c_size_t(0)
            t = time.perf_counter()
            L.check(L.lib().anx_find_all_matches_batch(m.h, arr, len(texts), C.byref(spc), C.byref(ms), C.byref(offs), C.byref(rows), C.byref(nrows), None))
            dt = time.perf_counter() - t
            L.lib().anx_matches_free(ms, offs, rows, None)
            if i_ >= 4:
                call_s.append(dt)
                best = dt if best is None else min(best, dt)
        thr_timed = _throttled() - thr0
        lat_ms, lat_n = L_.kernel_time("k_lattice")
        lm_ms, _lm_n = L_.kernel_time("k_lattice_lm")
        L_.kernel_timer(False)
        off, ma, ra = m.find_all_matches_arrays(texts, sp)
        # parity AND the CPU side of this configuration: sampled texts through the C oracle's search mode (oracle/anx_oracle_search.inc: the
        # reference's segmentation, lattice, k-best and bigram-LM rerank restated in C, pinned to the reference's 07xx tests and to the
        # Python twin by tests/test_oracle_search_c.py), one OpenMP task per text on this box's host cores
        om = O.OracleModel(alphabet_path=paths["alphabet"])
        om.read_lexicon(paths["eng"])
        for t_, f_ in lm:
            om.add_lm(t_, f_)
        om.build()
        osp = O.make_search_params(O.make_params(("abs", 3), ("abs", 2), 10, 0.25, 2.0), max_ngram=3)
        nchk = 8 * ncores
        idx = random.Random(5).sample(range(len(texts)), nchk)
        for i in idx:
            exp, _pairs = om.find_all_matches(texts[i], osp)
            got = ma[off[i]:off[i + 1]]
            if [(int(g_["begin"]), int(g_["end"])) for g_ in got] != [(e[1], e[2]) for e in exp]:
                raise RuntimeError(f"parity spot check failed: segmentation of text {i}")
            for g_, e in zip(got, exp):
                ev = e[5] or []
                rows = ra[int(g_["vb"]):int(g_["ve"])]
                if [int(v) for v in rows["vocab_id"]] != [v[0] for v in ev] or (ev and int(g_["selected"]) != e[4]) \
                        or any(abs(float(r["dist"]) - w[1]) > 1e-6 for r, w in zip(rows, ev)):
                    raise RuntimeError(f"parity spot check failed: text {i}, match {e[0]!r}")
        # the CPU baseline: a bounded sample of the same texts, calibrated on a short run
        cal = [texts[i] for i in idx[:2 * ncores]]
        t = time.perf_counter()
        om.find_all_matches_batch(cal, osp, nthreads=ncores)
        rate = sum(len(t_.encode("utf-8")) for t_ in cal) / max(time.perf_counter() - t, 1e-3)   # bytes/s
        nsamp = int(max(len(cal), min(len(texts), rate * 10.0 / (nbytes / len(texts)))))
        samp = random.Random(6).sample(texts, nsamp)
        t = time.perf_counter()
        rc_, _counts, tm_, _tr, tp_ = om.find_all_matches_batch(samp, osp, nthreads=ncores)
        dt_c = time.perf_counter() - t
        samp_bytes = sum(len(t_.encode("utf-8")) for t_ in samp)
        cpu4 = {"value": samp_bytes / 1e6 / dt_c, "unit": "MB/s", "cores": ncores, "kind": "port", "matches_per_s": tm_ / dt_c, "scored_pairs_per_s": tp_ / dt_c,
                "sample": f"{nsamp} of the {len(texts)} texts ({samp_bytes} bytes), C oracle's find_all_matches (oracle/anx_oracle_search.inc over oracle/anx_oracle.c), "
                          f"OpenMP, one task per text, {ncores} threads, {dt_c:.1f} s" + ("" if rc_ == 0 else " (a capacity did not hold for some text)")}
        del om
        return {"workload": "BASELINE.json configs[4], one GPU's share: 12.5 MB of synthetic running text (sentences of 5-25 perturbed words), max_ngram 3, bigram LM, anx_find_all_matches_batch",
                "MB_per_s": nbytes / 1e6 / best, "seconds": best, "cpu_baseline": cpu4, "median_MB_per_s": nbytes / 1e6 / sorted(call_s)[len(call_s) // 2], "calls": "4 untimed + 5 timed, best / median of the timed ones",
                "call_seconds": [round(x, 5) for x in call_s], "cgroup_throttled_periods_during_timed_calls": thr_timed,
                "matches": int(off[-1]), "variant_rows": int(ra.shape[0]),
                # k_lattice per call (all its launches): algorithmic bytes = the lattice input (16 B per arc: one arc per variant row, plus
                # one out-of-vocabulary / epsilon arc per match) + the chosen symbols out (8 B per match)
                "roofline": (lambda ms, nb: {"bound": "valu_issue", "kernel": "k_lattice", "avg_kernel_ms": ms, "algorithmic_bytes": nb, "compulsory_bytes": nb, "access_bytes": nb,
                                             "achieved": (nb / (ms * 1e-3) / 1e9 if ms > 0 else 0.0), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": (nb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else 0.0), "valu_issue_frac": None,
                                             "traffic": _profile_traffic("search", "k_lattice")[0], "traffic_source": _profile_traffic("search", "k_lattice")[1],
                                             "launches_per_call": lat_n / 5.0,
                                             "k_lattice_lm_ms": lm_ms / 5.0,
                                             "note": "ms = all k_lattice launches of one call (the four parts' launches overlap on the device: each takes longer than alone; "
                                                     "12.5 MB as ONE part: profiles/r*_search_single_part_kernel_trace.md); two stretches per wave, K <= 250 pops of a "
                                                     "group-wide minimum each, ~17 vector instructions per pop: bound by vector-instruction issue (DESIGN.md section 5 K6); "
                                                     "k_lattice_lm (LM sums, rerank, chosen path): dependent loads, latency"})(
                    lat_ms / 5.0, int(ra.shape[0]) * 16 + int(off[-1]) * (16 + 8)),
                "parity": f"ok ({nchk} texts = {8 * nchk} sentences vs the C oracle's search mode, itself pinned to the twin)"}

    for name, fn in (("nld_len16_d2", nld_len16_d2), ("configs2_nld_d3_confusables", configs2), ("configs3_share", configs3_share),
                     ("configs4_share_search", configs4_share)):
        if not args.extras or name in args.extras.split(","):
            guarded(name, fn)
    return out


def single_process(args):
    """--single-process: ONE process, --gpus N replicas of the lexicon behind the C ABI (anx_model_to_devices: one host thread and
    one stream per device, contiguous input ranges, rows concatenated in input order).  Weak scaling like the rank form: every
    replica gets --queries inputs.  No torch.distributed and no collective: the rows are host-consumed."""
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the variant-query path has no CPU fallback")
    import analiticcl_amd as A
    from analiticcl_amd import synth
    n = args.gpus
    ndev = torch.cuda.device_count()
    devices = [0] * n if args.replicas_on_one_gpu else list(range(n))
    if max(devices) >= ndev:
        raise SystemExit(f"bench.py --single-process --gpus {n}: only {ndev} device(s) visible (use --replicas-on-one-gpu for a dry run)")
    paths = synth.materialize_golden(os.path.join(tempfile.gettempdir(), f"anx_bench_data_{os.getuid()}_sp"))
    model = A.VariantModel(paths["alphabet"], A.Weights(), devices=devices)
    model.read_lexicon(paths[args.lexicon])
    model.build()
    words = synth.load_lexicon_words(paths[args.lexicon])
    per = [synth.make_queries(words, args.queries, max_len=args.max_len, seed=synth.SEED + r) for r in range(n)]  # the rank form's shards
    queries = [q for part in per for q in part]
    params = A.SearchParameters(max_anagram_distance=args.anagram_distance, max_edit_distance=args.edit_distance, max_matches=10,
                                score_threshold=0.25, cutoff_threshold=2.0)
    A.set_switch("ANX_SHARD_MIN", 1024)
    batches = [model.encode_batch(queries, params) for _ in range(2)]
    shards = batches[0].shards()
    assert len(shards) == n, shards
    inflight = []
    kernel_ms = {"ms_scan_kernel": 0.0, "ms_filter_score_kernel": 0.0, "ms_total": 0.0}
    done = [0]

    def finish(collect):
        b = batches[inflight.pop(0)]
        b.wait()
        if collect:
            st_ = b.stats()
            for k in kernel_ms:
                kernel_ms[k] += st_[k]
            done[0] += 1

    def step(k, collect):
        i = k & 1
        if i in inflight:
            finish(collect)
        batches[i].run_async(0)
        inflight.append(i)

    def sync_all():
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)

    for k in range(args.warmup):
        step(k, False)
    while inflight:
        finish(False)
    sync_all()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, True)
    while inflight:
        finish(True)
    sync_all()
    elapsed = time.perf_counter() - t0
    st = batches[0].stats()
    # shard == whole: replica 0's rows equal those of the same queries run on a one-replica batch
    check = None
    if not args.timed_only:
        off, vid, dist_, _f = batches[0].fetch_arrays()
        A.set_switch("ANX_SHARD_MIN", 1 << 30)
        b1 = model.encode_batch(per[-1], params)   # the LAST replica's inputs, on one replica
        A.set_switch("ANX_SHARD_MIN", 1024)
        b1.run()
        o1, v1, d1, _f1 = b1.fetch_arrays()
        b1.free()
        lo = (n - 1) * args.queries
        import numpy as np
        same = np.array_equal(o1, off[lo:] - off[lo]) and np.array_equal(v1, vid[off[lo]:]) and np.array_equal(d1, dist_[off[lo]:])
        check = "ok" if same else "rows of the last shard differ from a one-replica run of the same inputs"
    steps = max(args.steps, 1)
    scan_ms, fs_ms = kernel_ms["ms_scan_kernel"] / steps, kernel_ms["ms_filter_score_kernel"] / steps
    # per-launch figures of ONE replica (the slowest): the roofline is a per-kernel quantity
    st1 = dict(st)
    for k in ("n_queries", "n_pairs", "n_class_tests", "n_results", "n_scan_blocks", "n_pair_slots", "n_survivors", "n_selected", "n_prefiltered_in_scan"):
        st1[k] = st[k] / n
    st1["n_tests_kind"] = [x / n for x in st["n_tests_kind"]]
    roofline = roofline_of(args, model, per[0], st1, scan_ms, fs_ms, kernel_ms["ms_total"] / steps)
    ncores = usable_cores()
    cpu = cpu_baseline_of(args, paths, per[0], ncores, 15.0 if n == 1 else 6.0) if (args.cpu_sample != 0 and not args.timed_only) else None
    out = {
        "metric": baseline_metric(), "value": st["n_pairs"] * args.steps / elapsed, "unit": "pairs/s",
        "queries_per_s": st["n_queries"] * args.steps / elapsed, "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": f"{args.lexicon}.aspell lexicon + simple.alphabet, {args.queries} synthetic queries len<={args.max_len} per GPU, "
                               f"k={args.anagram_distance} d={args.edit_distance} n=10 score-threshold 0.25 cutoff 2.0",
                   "queries_per_gpu": args.queries, "lexicon_entries": model.num_instances(), "anagram_classes": model.num_classes(),
                   "pairs_per_query": st["n_pairs"] / max(st["n_queries"], 1),
                   "parallelism": f"query-sharded x{n}, ONE process: anx_model_to_devices({devices}), one host thread + stream per replica, length-partitioned split of every call, rows back in input order on the host (no collective)"},
        "pipelining": "2 resident copies of the sharded batch, anx_batch_run_async on every replica's own stream, waited for a step later",
        "shards": shards, "shard_check": check, "process_group": None, "roofline": roofline, "cpu_baseline": cpu,
    }
    for b in batches:
        b.free()
    print(json.dumps(out))


def strong_scaling_job(args):
    """--strong (with --single-process --gpus N): BASELINE configs[3]'s job SHAPE as a strong-scaling measurement behind the C ABI -- ONE
    process, the merged 1 M-entry lexicon replicated on N devices (anx_model_to_devices), `--strong-queries` length-bucketed queries of
    4-32 symbols split by the length-partitioned policy, every replica's top-k records gathered into one buffer on device 0
    (anx_batch_gather_compact: in place / hipMemcpyPeerAsync over xGMI).  The same job is first run on ONE replica: the ratio of the two
    times is the speed-up the driver can turn into an efficiency.  Prints one JSON line."""
    import numpy as np
    import torch
    import analiticcl_amd as A
    from analiticcl_amd import synth
    n = args.gpus
    ndev = torch.cuda.device_count()
    devices = [0] * n if args.replicas_on_one_gpu else list(range(n))
    if max(devices) >= ndev:
        raise SystemExit(f"--strong --gpus {n}: only {ndev} device(s) visible (use --replicas-on-one-gpu for a dry run)")
    paths = synth.materialize_golden(os.path.join(tempfile.gettempdir(), f"anx_bench_data_{os.getuid()}_strong"))
    words = list(dict.fromkeys(synth.load_lexicon_words(paths["eng"]) + synth.load_lexicon_words(paths["nld"])))
    lex = synth.make_lexicon(words, args.strong_entries, seed=11)
    path = os.path.join(tempfile.gettempdir(), f"anx_bench_strong_{os.getuid()}.lexicon")
    with open(path, "w", encoding="utf-8") as f:
        f.write("\n".join(lex) + "\n")
    qs = synth.make_queries(lex, args.strong_queries, max_len=32, min_len=4, seed=6)
    p = A.SearchParameters(max_anagram_distance=3, max_edit_distance=2, max_matches=10, score_threshold=0.25, cutoff_threshold=2.0)
    # peer access between device 0 (the gather's destination) and every other device in use
    peers = {f"{d}->0": bool(torch.cuda.can_device_access_peer(d, 0)) for d in sorted(set(devices)) if d != 0}

    def timed(model, reps):
        b = model.encode_batch(qs, p)
        b.run()
        b.run()   # (the second call's split has learned from the first one's shard times)
        t = time.perf_counter()
        for _ in range(reps):
            b.run()
        dt = (time.perf_counter() - t) / reps
        return b, dt
    one = A.VariantModel(paths["alphabet"], A.Weights(), device=devices[0])
    one.read_lexicon(path)
    one.build()
    b1, t1 = timed(one, 3)
    o1, v1, d1, _f1 = b1.fetch_arrays()
    ref = (o1.copy(), v1.copy(), d1.copy())
    st1 = b1.stats()
    b1.free()
    del one
    A.set_switch("ANX_SHARD_MIN", 1024)
    many = A.VariantModel(paths["alphabet"], A.Weights(), devices=devices)
    many.read_lexicon(path)
    many.build()
    bn, tn = timed(many, 3)
    shards = bn.shards()
    on, vn, dn, _fn = bn.fetch_arrays()
    same = bool(np.array_equal(on, ref[0]) and np.array_equal(vn, ref[1]) and np.array_equal(dn, ref[2]))
    # the gather: every shard's compact records into one buffer on device 0
    cap = int(16 * (len(qs) + 64 * n) + 16 * int(on[-1]) + 4096 * n + (1 << 20))
    with torch.cuda.device(devices[0]):
        buf = torch.empty(cap, dtype=torch.uint8, device=f"cuda:{devices[0]}")
        torch.cuda.synchronize()
        t = time.perf_counter()
        offs, used = bn.gather_compact(devices[0], buf.data_ptr(), buf.numel())
        gather_s = time.perf_counter() - t
    bn.free()
    os.unlink(path)
    print(json.dumps({"what": "strong scaling behind the C ABI: the same job on 1 replica and on N (anx_model_to_devices + length-partitioned split + anx_batch_gather_compact)",
                      "workload": f"merged {len(lex)}-entry synthetic lexicon, {len(qs)} queries len 4-32, k=3 d=2 n=10 (BASELINE configs[3]'s shape)",
                      "devices": devices, "n_replicas": n, "ms_one_replica": t1 * 1e3, "ms_n_replicas": tn * 1e3, "speedup": t1 / tn,
                      "pairs_per_s_n_replicas": st1["n_pairs"] / tn, "rows_equal_one_replica_run": same,
                      "shards": [{"device": d_, "first_input": lo_, "inputs": c_} for d_, lo_, c_ in shards],
                      "gather": {"bytes": int(used), "seconds": gather_s, "GB_per_s": used / 1e9 / max(gather_s, 1e-9), "section_offsets": [int(x) for x in offs]},
                      "peer_access_to_device0": peers, "all_peers_reachable": all(peers.values()) if peers else None}))


def main():
    if os.environ.get("ANX_BENCH_WATCHDOG"):   # diagnosis: every thread's Python stack to stderr after N seconds, then exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["ANX_BENCH_WATCHDOG"]), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--queries", type=int, default=1_000_000)
    ap.add_argument("--max-len", type=int, default=16)
    ap.add_argument("--lexicon", default="eng", choices=["eng", "nld"])
    ap.add_argument("--cpu-sample", type=int, default=-1, help="queries timed on the CPU oracle (0 = skip)")
    ap.add_argument("--anagram-distance", type=int, default=3)
    ap.add_argument("--edit-distance", type=int, default=2)
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--force-gather", action="store_true", help="run the export + gather code at N=1 too (testing)")
    ap.add_argument("--timed-only", action="store_true", help="skip the passes after the timed region (two-stream overlap, end to end, CPU baseline): for rocprofv3 runs")
    ap.add_argument("--check-gather", action="store_true", default=True, help="rank 0: compare every rank's gathered export of the last step with that rank's fetch() (default whenever results are gathered)")
    ap.add_argument("--no-check-gather", dest="check_gather", action="store_false")
    ap.add_argument("--single-process", action="store_true", help="one process, --gpus N replicas behind the C ABI (anx_model_to_devices)")
    ap.add_argument("--replicas-on-one-gpu", action="store_true", help="with --single-process: all replicas on device 0")
    ap.add_argument("--strong", action="store_true", help="with --single-process: the strong-scaling job (configs[3]'s shape on 1 and on N replicas + the gather) instead of the weak-scaling steps")
    ap.add_argument("--strong-entries", type=int, default=1_000_000)
    ap.add_argument("--strong-queries", type=int, default=4_000_000)
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the single-process strong-scaling leg rank 0 runs after the timed region")
    ap.add_argument("--ranks-on-one-gpu", type=int, default=0, metavar="N", help="N ranks, all on device 0 (dry run of the N-rank control flow)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend of the result gather")
    ap.add_argument("--no-extras", action="store_true", help="skip the per-config numbers measured after the timed region")
    ap.add_argument("--no-overlap", action="store_true", help="ANX_RUN_OVERLAP=0 for the whole run: every kernel alone on the GPU (rocprofv3 passes: clean per-kernel durations)")
    ap.add_argument("--extras", default="", help="comma-separated names of the extra configurations to run (default: all)")
    ap.add_argument("--preroll-s", type=float, default=2.5, help="seconds of untimed steady-state steps before the timed region (lets an external "
                    "GPU-utilisation sampler see the device busy; 0 with --timed-only)")
    ap.add_argument("--spot-check", type=int, default=256, help="queries of the timed batch (and of every extra configuration) checked against the oracle after the timed region")
    args = ap.parse_args()
    if args.ranks_on_one_gpu:
        args.gpus = args.ranks_on_one_gpu
    if args.single_process:
        return strong_scaling_job(args) if args.strong else single_process(args)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # (also --ranks-on-one-gpu N)
        # `python bench.py --gpus N` without a launcher: start the N ranks as a fresh child (nothing here has touched the
        # GPU yet -- a process that has initialised HIP must not be replaced) and relay its output and exit code
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (python -m torch.distributed.run "
                         f"--nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...) or run `python bench.py --gpus {args.gpus}` alone")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the variant-query path has no CPU fallback")
    device = 0 if args.ranks_on_one_gpu else local_rank
    torch.cuda.set_device(device)
    use_dist = world > 1 or "WORLD_SIZE" in os.environ  # under a launcher the process group is set up at N=1 too (RCCL init + the
    if use_dist:                                          # collectives below run with one rank: what a one-GPU box can exercise)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")
    stage_host = use_dist and args.backend == "gloo"  # gloo moves host memory: the exported records are staged through pinned buffers

    import analiticcl_amd as A
    from analiticcl_amd import synth

    paths = synth.materialize_golden(os.path.join(tempfile.gettempdir(), f"anx_bench_data_{os.getuid()}_{rank}"))
    model = A.VariantModel(paths["alphabet"], A.Weights(), device=device)
    if world > 1:
        # N concurrent index builds would oversubscribe the host (the GPU boxes grant 16 CPUs): rank 0 builds and saves the image
        # (anx_model_save_index), the others load it
        image = os.path.join(tempfile.gettempdir(), f"anx_bench_index_{os.getuid()}_{os.environ.get('MASTER_PORT', '0')}_{args.lexicon}.idx")
        if rank == 0:
            model.read_lexicon(paths[args.lexicon])
            model.build()
            model.save_index(image)
        dist.barrier()
        if rank != 0:
            model.load_index(image)
        dist.barrier()
        if rank == 0:
            os.unlink(image)
    else:
        model.read_lexicon(paths[args.lexicon])
        model.build()
    words = synth.load_lexicon_words(paths[args.lexicon])
    queries = synth.make_queries(words, args.queries, max_len=args.max_len, seed=synth.SEED + rank)
    params = A.SearchParameters(max_anagram_distance=args.anagram_distance, max_edit_distance=args.edit_distance, max_matches=10,
                                score_threshold=0.25, cutoff_threshold=2.0)
    if args.no_overlap:
        A.set_switch("ANX_RUN_OVERLAP", "0")
    if os.environ.get("ANX_BENCH_E2E_FIRST"):  # diagnosis: the end-to-end section in a fresh process state
        e_ = e2e_of(args, model, queries, params, torch.cuda.current_stream().cuda_stream, torch)
        e_.pop("pipelined_last", None)
        sys.stderr.write("[bench] e2e first: " + json.dumps({k: v for k, v in e_.items() if k != "what"}) + "\n")
    t_enc = time.time()
    batch = model.encode_batch(queries, params)  # encode + H2D, outside the timed region
    t_enc = time.time() - t_enc
    # Two resident copies of the batch alternate, so that step i+1 is enqueued while step i still runs: a run has no host round
    # trip inside (anx_batch_run_async), the one read-back at its end is waited for a step later, and the stream never idles
    # between steps.  ONE stream: the kernels of consecutive steps do not overlap, the per-kernel HIP-event times stay clean.
    batches = [batch, model.encode_batch(queries, params)]
    stride = 11  # max_matches + 1 records per query (crop tie rule can return max_matches + 1)
    stream = torch.cuda.current_stream()
    do_gather = (world > 1 or args.force_gather) and not args.no_gather
    # The only exchange of the path: top-k records -> rank 0 (RCCL point-to-point over xGMI).  Double-buffered
    # and asynchronous, so the gather of step i overlaps the scan/score kernels of step i+1.
    # Compact records (offsets + the rows in use: 74 MB per million queries of this workload instead of 176 MB at a
    # fixed stride), sizes exchanged one step ahead of the payloads: analiticcl_amd/shard.py CompactGather.
    from analiticcl_amd import shard
    cap = shard.compact_capacity(args.queries, stride + 5)
    gather = shard.CompactGather(cap, "cpu" if stage_host else "cuda", rank, world) if do_gather else None
    dev_stage = [torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in range(2)] if (do_gather and stage_host) else None
    if do_gather and stage_host:
        gather.send = [t.pin_memory() for t in gather.send]
    step_no = [0]
    gather_bytes = [0]
    gather_error = [None]
    stage_ms = {"ms_scan": 0.0, "ms_group": 0.0, "ms_score": 0.0, "ms_rank": 0.0, "ms_total": 0.0}
    kernel_ms = {"ms_scan_kernel": 0.0, "ms_filter_score_kernel": 0.0}
    finished = [0]
    inflight = []  # (batch index, hip stream) in launch order

    def finish_oldest(collect):
        i, hs = inflight.pop(0)
        b = batches[i]
        b.wait()
        if collect:
            st = b.stats()  # HIP-event times recorded by the library on the launch stream for this run
            for k in stage_ms:
                stage_ms[k] += st[k]
            for k in kernel_ms:
                kernel_ms[k] += st[k]
            finished[0] += 1
        if do_gather and gather_error[0] is None:
            try:
                j = step_no[0] & 1
                buf = gather.acquire(j)        # the stream waits for the transfer that last used this buffer
                if stage_host:                 # gloo: device export -> pinned host buffer, then the exchange
                    used = b.export_compact(dev_stage[j].data_ptr(), dev_stage[j].numel(), hs)
                    with torch.cuda.stream(stream):
                        buf[:used].copy_(dev_stage[j][:used], non_blocking=True)
                    stream.synchronize()
                else:
                    used = b.export_compact(buf.data_ptr(), buf.numel(), hs)
                gather.submit(j, used)
                gather_bytes[0] = used
                step_no[0] += 1
            except Exception as e:  # noqa: BLE001
                if world > 1:  # the other ranks keep posting collectives: a rank that stops would leave them mismatched
                    raise
                gather_error[0] = repr(e)[:200]  # N=1 (--force-gather): keep the compute measurement, the JSON line says what happened
                sys.stderr.write(f"[bench] rank {rank}: result gather disabled: {gather_error[0]}\n")

    def step(k, hs, collect):
        i = k & 1
        if any(x[0] == i for x in inflight):
            finish_oldest(collect)     # the previous run of this copy (two steps ago)
        batches[i].run_async(hs)
        inflight.append((i, hs))

    def drain(collect):
        while inflight:
            finish_oldest(collect)
        if do_gather and gather_error[0] is None:
            try:
                gather.flush()
            except Exception as e:  # noqa: BLE001
                if world > 1:
                    raise
                gather_error[0] = repr(e)[:200]
                sys.stderr.write(f"[bench] rank {rank}: result gather failed in flush: {gather_error[0]}\n")

    def barrier(collect=False):
        drain(collect)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k, stream.cuda_stream, False)
    barrier()
    # untimed pre-roll: the same steps for a couple of seconds, so that the device is visibly busy before the (sub-second) timed
    # region starts and runs at its steady-state clocks
    preroll_steps = 0
    if args.preroll_s > 0 and not args.timed_only:
        # The SAME number of steps on every rank: a step posts a collective (the size exchange of the result gather), and a loop that
        # each rank ends by its own clock left ranks with different counts -- one rank in the barrier, the others waiting for its
        # gather (seen as an intermittent hang of the 3-ranks-on-one-GPU job, 3 in 24 runs, once the steps had become shorter).
        # One chunk of 8 steps is timed, the slowest rank's time decides how many more chunks everybody runs.
        t_pre = time.perf_counter()
        for k in range(8):
            step(k, stream.cuda_stream, False)
        barrier()
        t8 = torch.tensor([time.perf_counter() - t_pre], dtype=torch.float64, device="cuda")
        if use_dist:
            dist.all_reduce(t8, op=dist.ReduceOp.MAX)
        t8 = max(float(t8.item()), 1e-4)
        nchunks = max(0, min(100000, int((args.preroll_s - t8) / t8)))
        for _ in range(nchunks):
            for k in range(8):
                step(k, stream.cuda_stream, False)
        preroll_steps = 8 * (1 + nchunks)
        barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, stream.cuda_stream, True)
    barrier(True)
    elapsed = time.perf_counter() - t0
    assert finished[0] == args.steps
    sum_scan_kernel_ms, sum_fs_kernel_ms = kernel_ms["ms_scan_kernel"], kernel_ms["ms_filter_score_kernel"]
    # The timed steps overlap (anx_batch_run_async alternates between two streams of the library: the scan of one step runs under
    # the tail of the previous one), so the HIP-event times of their kernels include the time they share the GPU.  The roofline
    # uses the kernels' OWN durations: the same steps once more with ANX_RUN_OVERLAP=0 (one stream, no kernel overlaps another) --
    # which is also how the rocprofv3 passes of profiles/ are taken.
    overlapped_kernel_ms = {"k_scan_bits": sum_scan_kernel_ms / max(args.steps, 1), "k_filter_score": sum_fs_kernel_ms / max(args.steps, 1)}
    serial_ms = None
    if args.no_overlap:
        serial_ms = elapsed / max(args.steps, 1) * 1e3
    elif A.lib().anx_debug_set_switch(b"ANX_RUN_OVERLAP", b"0") == 0:
        for k in range(2):
            step(k, stream.cuda_stream, False)
        barrier()
        for k_ in kernel_ms:
            kernel_ms[k_] = 0.0
        for k_ in stage_ms:
            stage_ms[k_] = 0.0
        finished[0] = 0
        t1 = time.perf_counter()
        for k in range(args.steps):
            step(k, stream.cuda_stream, True)
        barrier(True)
        serial_ms = (time.perf_counter() - t1) / max(args.steps, 1) * 1e3
        assert finished[0] == args.steps
        sum_scan_kernel_ms, sum_fs_kernel_ms = kernel_ms["ms_scan_kernel"], kernel_ms["ms_filter_score_kernel"]
        A.lib().anx_debug_set_switch(b"ANX_RUN_OVERLAP", None)
    overlapped = None  # (until round 3: the same steps on two caller streams; the library overlaps consecutive runs itself now)
    # one resident copy, one run at a time (anx_batch_run: launch, wait, launch ...): the like-for-like figure of round 1's records.
    # Measured here, while the GPU is still busy (after the CPU baseline its clocks have dropped).
    sync_ms = None
    if world == 1 and not do_gather and not args.timed_only:
        batches[0].run(stream.cuda_stream)
        t1 = time.perf_counter()
        nsync = max(4, args.steps)
        for _ in range(nsync):
            batches[0].run(stream.cuda_stream)
        sync_ms = (time.perf_counter() - t1) / nsync * 1e3
    # The step WITH the query encoder (normalize_to_alphabet, the count vector behind anahash,
    # the clamps): the raw input bytes are resident in HBM (one packed buffer, every string followed by a NUL byte), a step = device
    # encoder (anx_batch_encode_packed_device: no PCIe) + anx_batch_run_async, the run waited for a step later -- the encoder of
    # step i + 1 works while the GPU runs step i.  Never `value`: reported next to it.
    with_encode = None
    with_encode_rows = None
    if world == 1 and not do_gather and not args.timed_only:
        packed_ = ("\0".join(queries) + "\0").encode("utf-8")
        dev_blob = torch.frombuffer(bytearray(packed_), dtype=torch.uint8).cuda()
        live = []

        def enc_step():
            b_ = model.encode_packed_device(dev_blob.data_ptr(), dev_blob.numel(), len(queries), params)
            b_.run_async(stream.cuda_stream)
            live.append(b_)
            if len(live) > 1:
                o_ = live.pop(0)
                o_.wait()
                o_.free()

        def enc_drain(keep_last=False):
            last_ = None
            while live:
                o_ = live.pop(0)
                o_.wait()
                if keep_last and not live:
                    last_ = o_.fetch_arrays()   # the rows of the last fresh batch: checked against the oracle below
                o_.free()
            torch.cuda.synchronize()
            return last_
        for _ in range(3):
            enc_step()
        enc_drain()
        t1 = time.perf_counter()
        nenc = max(8, args.steps)
        for _ in range(nenc):
            enc_step()
        enc_drain()
        dt_ = (time.perf_counter() - t1) / nenc
        enc_step()   # one more fresh batch, outside the timed loop: its rows are checked against the oracle below
        with_encode_rows = enc_drain(keep_last=True)
        with_encode = {"ms_per_step": dt_ * 1e3, "queries_per_s": len(queries) / dt_, "pairs_per_s": batch.stats()["n_pairs"] / dt_, "steps": nenc,
                       "what": "raw packed input bytes resident in HBM -> device-side encoder (k_enc_strings, sort, k_enc_gather, tiles) -> scan -> score -> rank; "
                               "a fresh batch per step (anx_batch_encode_packed_device + anx_batch_run_async), the run of step i waited for and freed after "
                               "step i + 1 was enqueued"}
        del dev_blob
    st = batch.stats()
    per_rank_ms = [elapsed / max(args.steps, 1) * 1e3]
    if use_dist and world > 1:  # every rank's own time per step: a first real N-GPU run shows at once which rank is the slow one
        per_rank_ms = [None] * world
        dist.all_gather_object(per_rank_ms, elapsed / max(args.steps, 1) * 1e3)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    tot = torch.tensor([float(st["n_pairs"]), float(st["n_queries"]), float(st["n_class_tests"])],
                       dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(tmax.item())
    pairs, nq, tests = (float(x) for x in tot.tolist())

    gather_check = None
    if do_gather and args.check_gather and gather_error[0] is None:
        # every rank: a digest of its own fetch() (offsets, vocab ids, dist scores: the fields of a compact record); rank 0: the same
        # digest of every rank's export as it sits in the gather buffers of the last step.  (Both resident copies hold the same
        # queries, so the last step's export equals batches[0]'s rows.)
        import hashlib

        import numpy as np

        def digest(off, vid, dscore):
            h = hashlib.sha256()
            for a_ in (np.ascontiguousarray(off, dtype="<u4"), np.ascontiguousarray(vid, dtype="<u4"), np.ascontiguousarray(dscore, dtype="<f8")):
                h.update(a_.tobytes())
            return h.hexdigest()
        off_, vid_, dist_, _freq = batches[0].fetch_arrays()
        mine = digest(off_, vid_, dist_)
        digests = [mine]
        if use_dist:
            digests = [None] * world
            dist.all_gather_object(digests, mine)
        if rank == 0:
            last = (step_no[0] - 1) & 1
            bad = []
            for r, part in enumerate(gather.result(last)):
                raw = part.cpu().numpy().tobytes()
                o_ = np.frombuffer(raw, dtype="<u4", count=args.queries + 1)
                rows_ = np.frombuffer(raw, dtype=shard.TOPK_DTYPE, count=int(o_[args.queries]), offset=shard.compact_offsets_bytes(args.queries))
                if digest(o_, rows_["vocab_id"], rows_["dist_score"]) != digests[r]:
                    bad.append(r)
            gather_check = "ok" if not bad else f"ranks {bad} differ"
    if rank == 0:
        for k in stage_ms:
            stage_ms[k] /= max(args.steps, 1)
        scan_ms, fs_ms = sum_scan_kernel_ms / max(args.steps, 1), sum_fs_kernel_ms / max(args.steps, 1)
        n_classes = model.num_classes()
        roofline = roofline_of(args, model, queries, st, scan_ms, fs_ms, stage_ms["ms_total"])
        e2e = e2e_of(args, model, queries, params, stream.cuda_stream, torch) if (world == 1 and not args.timed_only) else None
        ncores = usable_cores()
        # parity of the TIMED batch itself: sampled queries of the rows the last timed step left on the device against the oracle
        # (ranked ids, f64 scores) -- after the timed region, never inside it
        parity = None
        by_batch_size = None
        if args.spot_check > 0 and not args.timed_only:
            from oracle import cwrap as O
            om_ = O.OracleModel(alphabet_path=paths["alphabet"])
            om_.read_lexicon(paths[args.lexicon])
            om_.build()
            op_ = O.make_params(("abs", args.anagram_distance), ("abs", args.edit_distance), 10, 0.25, 2.0)
            parity = _spot_check(model, om_, queries, batches[(args.steps - 1) & 1].fetch_arrays(), op_, min(args.spot_check, len(queries)))
            # the paths a caller uses, each against the ORACLE (not against the synchronous path): a fresh batch of the with_encode loop
            # (device-resident inputs) and a batch that went through anx_pipeline
            if with_encode is not None and with_encode_rows is not None:
                with_encode["parity"] = _spot_check(model, om_, queries, with_encode_rows, op_, min(args.spot_check, len(queries)))
            if e2e is not None and e2e.get("pipelined_last") is not None:
                off_, rows_ = e2e.pop("pipelined_last")
                e2e["pipelined_oracle_parity"] = _spot_check(model, om_, queries, (off_, rows_["vocab_id"], rows_["dist_score"], rows_["freq_score"]), op_,
                                                             min(args.spot_check, len(queries)))
            if world == 1 and not do_gather:
                by_batch_size = by_batch_size_of(model, om_, queries, params, op_, alphabet_path=paths["alphabet"], lexicon_path=paths[args.lexicon])
            del om_
        if e2e is not None:
            e2e.pop("pipelined_last", None)
        # the CPU baseline is reported from rank 0; at N > 1 on a shorter sample (the other ranks wait at the final barrier)
        cpu = cpu_baseline_of(args, paths, queries, ncores, 15.0 if world == 1 else 6.0) if (args.cpu_sample != 0 and not args.timed_only) else None
        # one resident copy, one run at a time (anx_batch_run: launch, wait, launch ...): the like-for-like figure of round 1's records
        extras = None
        if world == 1 and not args.timed_only:
            is_default = (args.lexicon, args.max_len, args.anagram_distance, args.edit_distance, args.queries) == ("eng", 16, 3, 2, 1_000_000)
            if is_default and not args.no_extras and not do_gather:
                for b_ in batches:
                    b_.free()
                extras = extra_configs(args, paths, device, ncores)
        out = {
            "metric": baseline_metric(),
            "value": pairs * args.steps / elapsed, "unit": "pairs/s",
            "pipelining": "2 resident copies of the batch alternate: anx_batch_run_async from ONE caller stream, each waited for a step later; the library runs "
                          "consecutive asynchronous runs on two streams of its own (the scan of one under the scoring tail / compaction / ranking of the other)",
            "serial_ms_per_step": serial_ms,
            "ms_per_step_with_encode": with_encode["ms_per_step"] if with_encode else None,
            "pairs_per_s_with_encode": with_encode["pairs_per_s"] if with_encode else None,   # the fresh-batch step: what a caller executes
            "queries_per_s_with_encode": with_encode["queries_per_s"] if with_encode else None,
            "with_encode": with_encode, "by_batch_size": by_batch_size,
            "kernels_ms_in_timed_region": overlapped_kernel_ms,
            "sync_single_copy_ms_per_step": sync_ms,
            "parity": parity, "preroll_steps": preroll_steps,
            "configs": extras,
            "queries_per_s": nq * args.steps / elapsed,
            # `value` counts the reference's scored pairs (every damerau_levenshtein call of gather_instances, src/lib.rs:1343);
            # 16 % of them fail its length test and are only counted, 2/3 of the rest are rejected by the exact prefilter:
            "dp_pairs_per_s": st["n_selected"] * world * args.steps / elapsed,       # pairs that ran the banded DL
            "lcs_pairs_per_s": st["n_survivors"] * world * args.steps / elapsed,     # ... and the LCS / prefix / suffix tail
            "e2e_queries_per_s": e2e["queries_per_s"] if e2e else None, "e2e": e2e,
            "overlapped": ({"ms_per_step": overlapped * 1e3, "pairs_per_s": pairs / overlapped, "queries_per_s": nq / overlapped,
                            "what": "the same steps with the two resident copies of the batch on two HIP streams (compaction + ranking of one run "
                                    "under the scan of the next), measured after the timed region"} if overlapped else None),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "ms_per_step_by_rank": per_rank_ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.lexicon}.aspell lexicon + simple.alphabet, {args.queries} synthetic queries "
                                   f"len<={args.max_len} per GPU, k={args.anagram_distance} d={args.edit_distance} n=10 score-threshold 0.25 cutoff 2.0"
                                   + (" (BASELINE.json configs[1])" if (args.lexicon, args.max_len, args.anagram_distance, args.edit_distance, args.queries) == ("eng", 16, 3, 2, 1_000_000) else ""),
                       "queries_per_gpu": args.queries, "lexicon_entries": model.num_instances(),
                       "anagram_classes": n_classes, "pairs_per_query": pairs / nq if nq else 0.0,
                       "class_tests_per_query": tests / nq if nq else 0.0,
                       "parallelism": f"query-sharded x{world}" + (f", RCCL gather of compact top-k records ({gather_bytes[0] / 1e6:.0f} MB per rank and step)" if do_gather and gather_error[0] is None else "")
                       + (f", result gather FAILED on rank 0: {gather_error[0]}" if gather_error[0] else "")},
            "stage_ms": {"scan": stage_ms["ms_scan"], "score": stage_ms["ms_score"], "compact": stage_ms["ms_group"], "rank": stage_ms["ms_rank"], "total": stage_ms["ms_total"]},
            "pair_slots": st["n_pair_slots"], "dl_pairs": st["n_selected"], "survivors": st["n_survivors"], "results": st["n_results"], "encode_upload_s": t_enc,
            "gather_error": gather_error[0], "gather_check": gather_check, "process_group": (args.backend if use_dist else None), "ranks_on_one_gpu": bool(args.ranks_on_one_gpu),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if world > 1 and not args.no_strong and not args.ranks_on_one_gpu:
            # The other way to use the node: ONE process with N replicas behind the C ABI, the strong-scaling job of BASELINE configs[3]'s
            # shape (the ranks idle at the barrier below meanwhile; a child process: this one has initialised the GPU).  Its JSON line goes
            # into this line; a failure is reported, never fatal.
            import subprocess
            try:
                r_ = subprocess.run([sys.executable, os.path.abspath(__file__), "--single-process", "--strong", "--gpus", str(world)],
                                    capture_output=True, text=True, timeout=1500)
                out["single_process_strong"] = json.loads(r_.stdout.strip().splitlines()[-1]) if r_.returncode == 0 else {"error": (r_.stderr or r_.stdout)[-600:]}
            except Exception as e_:  # noqa: BLE001
                out["single_process_strong"] = {"error": repr(e_)[:300]}
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
