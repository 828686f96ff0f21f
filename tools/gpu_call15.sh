#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/call15
mkdir -p $O
cd $R
for v in "X=1" "ANX_SIG_GROUPS=6"; do
  echo "== nld d3 $v" | tee -a $O/summary.txt
  env $v timeout 300 python tools/scan_probe.py nld 1000000 24 3 2>&1 | grep -E "^(default)" | tee -a $O/summary.txt
done
timeout 1700 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1; echo "pytest all rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest_all.log
timeout 300 python tools/search_bench.py 12.5 2>&1 | tail -3 | tee -a $O/summary.txt
