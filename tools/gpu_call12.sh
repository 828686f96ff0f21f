#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/call12
mkdir -p $O
cd $R
timeout 300 python tools/scan_probe.py 2>&1 | grep -E "^(default|score)" | tee -a $O/summary.txt
timeout 300 python tools/scan_probe.py nld 1000000 24 3 2>&1 | grep -E "^(default|score)" | tee -a $O/summary.txt
timeout 1700 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1; echo "pytest all rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest_all.log
