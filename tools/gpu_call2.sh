#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/call2
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -25 $O/pytest_gpu.log
for v in "default" "ANX_SCAN_WALK=flat" "ANX_SIG_GROUPS=7" "ANX_SIG_GROUPS=8" "ANX_SIG_GROUPS=5"; do
  n=$(echo $v | tr '=' '_')
  if [ "$v" = "default" ]; then timeout 300 python bench.py --cpu-sample 0 > $O/bench_$n.log 2>&1; else env $v timeout 300 python bench.py --cpu-sample 0 > $O/bench_$n.log 2>&1; fi
  tail -1 $O/bench_$n.log > $O/bench_$n.json
  python - "$n" <<'PY'
import json,sys
n=sys.argv[1]
try:
    j=json.load(open(f"gpurun_out/call2/bench_{n}.json")); print(n, "ms/step %.3f"%j["ms_per_step"], {k:round(v,3) for k,v in j["stage_ms"].items()}, {k:round(v,3) for k,v in j["roofline"]["kernels_ms"].items()}, "tests/q", round(j["config"]["class_tests_per_query"]), "tiles?", j["roofline"].get("scan_tests_by_planes"))
except Exception as e: print(n, "ERR", e)
PY
done 2>&1 | tee -a $O/summary.txt
for v in "default" "ANX_SCAN_WALK=flat" "ANX_SIG_GROUPS=7" "ANX_SIG_GROUPS=8"; do
  n=$(echo $v | tr '=' '_')
  if [ "$v" = "default" ]; then timeout 600 python tools/big_lexicon_bench.py > $O/big_$n.log 2>&1; else env $v timeout 600 python tools/big_lexicon_bench.py > $O/big_$n.log 2>&1; fi
  echo "big $n: $(grep -E 'encode|ms_scan|spot' $O/big_$n.log | tr '\n' ' ')" | tee -a $O/summary.txt
done
