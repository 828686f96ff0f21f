#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/call16
mkdir -p $O
cd $R
timeout 300 python tools/scan_probe.py 2>&1 | grep -E "^(default|scan dbg)" | tee -a $O/summary.txt
timeout 600 python tools/big_lexicon_bench.py 2>&1 | grep -E 'encode|ms_scan|spot' | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_encode.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q > $O/pytest_a.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest_a.log
