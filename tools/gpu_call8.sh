#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/call8
mkdir -p $O
cd $R
timeout 300 python tools/scan_probe.py > $O/scan_probe.log 2>&1; grep -E "^(default|scan|score|len)" $O/scan_probe.log | tee -a $O/summary.txt
timeout 300 python tools/scan_probe.py nld 1000000 24 3 > $O/scan_probe_nld.log 2>&1; grep -E "^(default|len)" $O/scan_probe_nld.log | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_encode.py -x -q > $O/pytest_a.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest_a.log
