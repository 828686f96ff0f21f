#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/call17
mkdir -p $O
cd $R
timeout 1700 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1; echo "pytest all rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest_all.log
bash tools/measure_round.sh > $O/measure.log 2>&1; tail -12 $O/measure.log | cut -c1-300
cd $R
ANX_SEARCH_TIMING=1 timeout 300 python tools/search_bench.py 12.5 2>&1 | grep -E "anx search|C ABI" | tail -12 | tee -a $O/summary.txt
timeout 300 python tools/e2e_timing.py 2>&1 | tail -7 | tee -a $O/summary.txt
