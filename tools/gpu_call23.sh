#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/call23
mkdir -p $O
cd $R
ANX_SEARCH_TIMING=1 timeout 300 python tools/search_bench.py 12.5 2>&1 | grep -E "anx search|C ABI" | tail -11 | tee -a $O/summary.txt
