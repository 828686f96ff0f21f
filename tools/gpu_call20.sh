#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/call22
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_config4.py -q -x > $O/pytest_s.log 2>&1; echo "pytest search rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest_s.log
ANX_SEARCH_TIMING=1 timeout 300 python tools/search_bench.py 12.5 2>&1 | grep -E "anx search|C ABI" | tail -6 | tee -a $O/summary.txt
