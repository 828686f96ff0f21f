#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/call3
mkdir -p $O
cd $R
timeout 300 python tools/scan_probe.py > $O/scan_probe.log 2>&1; cat $O/scan_probe.log | grep -v "^$" | tail -12
timeout 1700 python -m pytest tests -m gpu -q --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -40 $O/pytest_gpu.log
