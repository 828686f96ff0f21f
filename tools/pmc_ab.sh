#!/bin/bash
# SQ counters of the bench's kernels for one or more library builds: pmc_ab.sh name1 [name2 ...]   (name = build/libanx_<name>.so, "tree" = the in-tree library)
# One rocprofv3 --pmc pass per build (kernel trace only, as the pool requires); prints the per-kernel sums of k_scan_bits / k_filter_score.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/meas
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  if [ "$n" = tree ]; then unset ANX_LIB; else export ANX_LIB=$R/build/libanx_$n.so; fi
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" ${PMC_EXTRA:+"FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"}; do
  rm -rf $O/pmc_ab
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_ab -- python3 $R/bench.py --steps 2 --warmup 1 --timed-only --no-overlap --no-extras --cpu-sample 0 > $O/pmc_ab.log 2>&1
  python3 - "$n" $(find $O/pmc_ab -name "*counter_collection.csv" | head -1) <<'PY'
import csv, sys, collections
name, path = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"]
    short = "k_scan_adj" if "k_scan_adj" in k else "k_scan_bits" if "k_scan_bits" in k else "k_filter_score" if "k_filter_score" in k else None
    if not short: continue
    acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES" or r["Counter_Name"] == "SQ_INSTS_LDS": cnt[short] += 1
for k, d in acc.items():
    n = max(cnt[k], 1)
    print(name, k, "dispatches", n, {c: round(v / n / 1e6, 2) for c, v in d.items()}, "(millions per dispatch)")
PY
  done
done
