"""gpurun_out/meas/* (written by tools/measure_round.sh on the GPU box) -> profiles/<round>_final_*.{md,json}
usage: collect_profiles.py [round tag, default r02]"""
import glob, hashlib, json, os, re, shutil, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r02"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M = os.path.join(R, "gpurun_out", "meas")
pmc = open(os.path.join(M, "pmc.md")).read()
def blocks():
    return re.findall(r"## (k_[a-z_0-9]+):.*?\n(.*?)\n\n", pmc + "\n\n", re.S)
def val(blk, c):
    m = re.search(c + r"\s+mean/dispatch\s+([\d.]+)", blk)
    return float(m.group(1)) if m else None
lines = []
kern = {}
for name, blk in blocks():
    f, w, va, g, vi, si = (val(blk, c) for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_INSTS_SALU"))
    if f is not None and w is not None:
        kern[name] = {"fetch_kb": f, "write_kb": w, "traffic_bytes": int((2 * f + w) * 1024)}
    if va and g and name in ("k_scan_adj", "k_scan_bits", "k_filter_score", "k_rank", "k_compact", "k_compact_grouped"):
        lines.append(f"# {name}: VALU-active = SQ_ACTIVE_INST_VALU*4 / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs) = {va*4/(g/8*1024)*100:.0f} %; "
                     f"{vi/1e6:.0f} M VALU + {si/1e6:.0f} M SALU wave-instructions; HBM traffic 2*{f/1024:.0f} MiB + {w/1024:.0f} MiB = {(2*f+w)*1024/1e9:.2f} GB per launch")
hdr = (f"# Round {int(TAG[1:])} final -- rocprofv3 --pmc passes (separate runs, --kernel-trace only), python3 bench.py --steps 2 --warmup 1 --timed-only --no-overlap (every kernel alone on the GPU)\n"
       "# recipe: tools/measure_round.sh; passes: {SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU} "
       "{SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE} {FETCH_SIZE} {WRITE_SIZE}\n"
       "# FETCH_SIZE / WRITE_SIZE are in KB; GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles.\n" + "\n".join(lines) + "\n\n")
open(os.path.join(R, "profiles", f"{TAG}_final_pmc.md"), "w").write(hdr + pmc)
sha = hashlib.sha256()
for f in sorted(glob.glob(os.path.join(R, "analiticcl_amd", "csrc", "*.hip")) + glob.glob(os.path.join(R, "analiticcl_amd", "csrc", "*.hpp"))):
    sha.update(open(f, "rb").read())
# the other configurations: HBM bytes per launch of every kernel of their PMC passes (tools/measure_configs.sh), keyed as bench.py asks
cfg = {}
for name in ("conf", "big", "search"):
    src = os.path.join(M, f"{name}_pmc.md")
    if not os.path.exists(src):
        continue
    body = open(src).read()
    ent = {}
    for kname, blk in re.findall(r"## (k_[a-z_0-9]+):.*?\n(.*?)\n\n", body + "\n\n", re.S):
        f, w = val(blk, "FETCH_SIZE"), val(blk, "WRITE_SIZE")
        if f is not None and w is not None:
            ent[kname] = {"fetch_kb": f, "write_kb": w, "traffic_bytes": int((2 * f + w) * 1024)}
    cfg[name] = ent
json.dump({"kernel_src_sha256": sha.hexdigest(), "source": f"profiles/{TAG}_final_pmc.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py configs[1], 1M queries); configs: profiles/{TAG}_{{conf,big,search}}_pmc.md",
           "note": "bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KB * 1024; the x2 on FETCH_SIZE is the gfx950 correction of MI355X_MICROARCH.md (calibrated there for 16 B/lane streams; narrower loads make the read side an upper bound)",
           "kernels": kern, "configs": cfg}, open(os.path.join(R, "profiles", f"{TAG}_pmc_traffic.json"), "w"), indent=1)
kt = open(os.path.join(M, "kernel_trace.md")).read()
open(os.path.join(R, "profiles", f"{TAG}_final_kernel_trace.md"), "w").write(
    f"# Round {int(TAG[1:])} final -- rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --timed-only --no-overlap (7 runs of the pipeline, every kernel alone on the GPU; tools/measure_round.sh)\n"
    "# config 2: eng.aspell, 1M queries len<=16, k=3 d=2 n=10; summarised from the rocpd database by profiles/summarize_rocpd.py\n\n" + kt)
shutil.copy(os.path.join(M, "bench_default.json"), os.path.join(R, "profiles", f"{TAG}_bench.json"))
print("\n".join(lines))

# the other configurations (tools/measure_configs.sh): kernel traces and PMC summaries as they are, with a header
for name, what in (("conf", "BASELINE configs[2]: tools/conf_probe.py 1000000 (nld.aspell, 1 M queries len <= 24, k = 3, d = 3, 10 confusable patterns; the model without patterns runs second)"),
                   ("big", "BASELINE configs[3], one GPU's share: tools/big_lexicon_bench.py 1000000 1250000 nocheck (merged 1 M-entry lexicon, 1.25 M length-bucketed queries len 4-32)"),
                   ("search", "BASELINE configs[4], one GPU's share: tools/search_bench.py 12.5 (12.5 MB of running text, max_ngram 3, bigram LM)")):
    for kind, hdr2 in (("kernel_trace", "rocprofv3 --kernel-trace --stats"), ("pmc", "rocprofv3 --pmc passes (separate runs, --kernel-trace only); mean per dispatch; FETCH_SIZE / WRITE_SIZE in KB")):
        src = os.path.join(M, f"{name}_{kind}.md")
        if os.path.exists(src) and os.path.getsize(src) > 0:
            body = open(src).read()
            if kind == "pmc":   # lanes active per vector instruction of the kernels that matter
                for kname, blk in re.findall(r"## (k_[a-z_0-9]+):.*?\n(.*?)\n\n", body + "\n\n", re.S):
                    tc, vi = val(blk, "SQ_THREAD_CYCLES_VALU"), val(blk, "SQ_INSTS_VALU")
                    if tc and vi and kname in ("k_conf_script", "k_lattice", "k_scan_adj", "k_scan_bits", "k_filter_score"):
                        body = f"# {kname}: SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU = {tc / vi:.1f} (lanes active per vector instruction; a kernel with every lane busy reads ~64)\n" + body
            open(os.path.join(R, "profiles", f"{TAG}_{name}_{kind}.md"), "w").write(
                f"# Round {int(TAG[1:])} -- {hdr2} -- {what}; ANX_RUN_OVERLAP=0 (tools/measure_configs.sh)\n\n" + body)

# the single-part search trace (tools/trace_cmd.sh lat1 with ANX_SEARCH_PARTS=1): k_lattice alone on the device
src = os.path.join(M, "trace_lat1.md")
if os.path.exists(src) and os.path.getsize(src) > 0:
    open(os.path.join(R, "profiles", f"{TAG}_search_single_part_kernel_trace.md"), "w").write(
        f"# Round {int(TAG[1:])} -- rocprofv3 --kernel-trace --stats -- tools/search_bench.py 12.5 with ANX_SEARCH_PARTS=1 (the whole 12.5 MB as ONE part: every kernel of the call alone "
        "on the device; k_lattice: total ms / calls = ms per 12.5 MB call, the bench makes 5 calls incl. the Python-level one)\n\n" + open(src).read())
