#!/usr/bin/env python3
"""gpurun_out/fresh_final + gpurun_out/small_final (tools/final_measure.sh) -> profiles/<tag>_with_encode_timeline.md, profiles/<tag>_small_call_trace.md.
usage: collect_extras.py [tag, default r06]"""
import json, os, re, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(R, "gpurun_out", "fresh_final")
S = os.path.join(R, "gpurun_out", "small_final")
ms = {m: json.loads(open(os.path.join(F, m + ".json")).read().strip().splitlines()[-1])["ms_per_step"] for m in ("loop", "encode", "run")}
bench = json.loads(open(os.path.join(R, "gpurun_out", "meas", "bench_default.json")).read().strip().splitlines()[-1])
tl = open(os.path.join(F, "timeline_loop.md")).read()
kt = open(os.path.join(R, "profiles", f"{TAG}_final_kernel_trace.md")).read()
alone = {}
for m in re.finditer(r"^\| (k_[a-z_0-9]+) \| \d+ \| [\d.]+ \| ([\d.]+) \|", kt, re.M):
    alone.setdefault(m.group(1), float(m.group(2)))
rows = []
for line in tl.splitlines():
    if not (line.startswith("| k_") or line.startswith("| __amd") or line.startswith("| void")):
        continue
    name, streams, launches, al, inl, stretch, delay, total = [x.strip() for x in line.strip("|").split("|")]
    src = "encoder alone (this tool)"
    if streams.startswith("2"):
        if name in alone:
            al, src, stretch = f"{alone[name]:.1f}", f"run alone (profiles/{TAG}_final_kernel_trace.md)", f"{float(inl) / alone[name]:.2f}"
        else:
            src = "first runs alone (this tool, cold clocks)"
    elif "," in streams:
        al, stretch, src = "-", "-", "-"
    rows.append(f"| {name} | {streams} | {launches} | {al} | {inl} | {stretch} | {delay} | {total} | {src} |")
hdr = f"""# Round {int(TAG[1:])} (final tree) -- the fresh-batch step (encode + run, what a caller executes) taken apart: rocprofv3 --kernel-trace of tools/fresh_batch.py loop
# (the with_encode loop of bench.py: raw packed inputs resident in HBM -> anx_batch_encode_packed_device -> anx_batch_run_async, a fresh batch per step,
# the run of step i waited for and freed after step i + 1 was enqueued), next to the same kernels alone (encoder kernels: the encode-only loop's trace;
# run kernels: profiles/{TAG}_final_kernel_trace.md, every kernel alone on the GPU).  tools/final_measure.sh (fresh_measure.sh + timeline.py) + tools/collect_extras.py;
# 12 traced steps, the last 75 % summarised.  Stream 1 = the encoder's high-priority stream, streams 2 / 3 = the library's two run streams.
#
# untraced, ms per 1 M-query step (this tool's short loops): loop {ms['loop']:.2f} (start of round 6: 3.79; round 5's bench line: 3.56; before the rebuild of k_filter_score: 2.71),
# encoder alone {ms['encode']:.2f}, first runs alone {ms['run']:.2f}; bench.py after its pre-roll: {bench['ms_per_step']:.2f} resident re-run, {bench['ms_per_step_with_encode']:.2f} with the encoder.
# start of round 6 (gpurun_out/fresh0, same tool): the loop's kernels did not overlap AT ALL -- the encoder of step i + 1 started 0.36 ms after the last kernel of run i,
# because batch_free(i - 1) called hipHostFree, which waits for the device.  Now the encoder runs under the previous run and the step is the run plus what the
# encoder's kernels take from it.

| kernel | stream | launches | alone avg us | in the loop avg us | stretch | avg wait behind its stream predecessor us | total in the loop ms | 'alone' from |
|---|---|---|---|---|---|---|---|---|
"""
open(os.path.join(R, "profiles", f"{TAG}_with_encode_timeline.md"), "w").write(hdr + "\n".join(rows) + "\n")


def call(n):
    out = []
    for line in open(os.path.join(S, f"call_{n}.txt")):
        p = line.split()
        if len(p) < 6 or float(p[0]) < 0:
            continue
        out.append(f"| {p[0]} | {p[1]} | {p[2]} | {p[4]} | {p[5][2:]} |")
    return "\n".join(out)


chk = open(os.path.join(S, "check.txt")).read()
sizes = re.search(r"\{'1': ([\d.]+), '64': ([\d.]+), '1000': ([\d.]+)", chk)
thr = {int(m.group(1)): int(m.group(2)) / 1e6 for m in re.finditer(r'"threads": (\d+), "n": 1000, "calls_per_thread": \d+, "queries_per_s": (\d+)', chk)}
txt = f"""# Round {int(TAG[1:])} (final tree) -- the small call (analiticcl_amd/csrc/small_path.hpp): rocprofv3 --kernel-trace of tools/small_trace.py N (anx_find_variants_batch, N inputs,
# eng.aspell, k=3 d=2 n=10), the kernels of ONE call in stream order (tools/dump_last_call.py; tools/final_measure.sh + tools/collect_extras.py).  Times in us from the start of k_enc_strings.
# Host to host, untraced (tools/small_check.sh): {sizes.group(1)} / {sizes.group(2)} / {sizes.group(3)} us for N = 1 / 64 / 1000 (before the small path: 561 / 663 / 847; first version of the path: 66 / 102 / 178);
# native host threads x 1000 inputs on one model (tools/small_threads.cpp): {' / '.join(f'{thr[t]:.1f}' for t in sorted(thr))} M queries/s with {' / '.join(str(t) for t in sorted(thr))} threads.
# Nine launches, one host wait; the gaps in front of the last kernels of the N = 1 chain are the host still enqueueing (a launch costs it ~4 us).

## N = 1000
| start | end | us | kernel | grid (blocks x threads) |
|---|---|---|---|---|
{call(1000)}

## N = 64
| start | end | us | kernel | grid (blocks x threads) |
|---|---|---|---|---|
{call(64)}

## N = 1
| start | end | us | kernel | grid (blocks x threads) |
|---|---|---|---|---|
{call(1)}

## the scan before / after a query's list was shared out over several waves (first half of round 6)
| N | k_scan_small before (one wave per query) | after |
|---|---|---|
| 64 | 94.2 us (the longest list: one wave waiting for its own loads, chunk after chunk) | 10.5 us |
| 1000 | 116.7 us | 32.9 us |
"""
open(os.path.join(R, "profiles", f"{TAG}_small_call_trace.md"), "w").write(txt)
print("written")
