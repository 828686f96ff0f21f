#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/call19
mkdir -p $O
cd $R
for v in "X=1" "ANX_SCAN_HITLIST=1" "X=2"; do
  echo "== $v" | tee -a $O/summary.txt
  env $v timeout 300 python tools/scan_probe.py 2>&1 | grep -E "^(default|scan dbg=8)" | tee -a $O/summary.txt
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_encode.py tests/test_gpu_config3.py -x -q > $O/pytest_a.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest_a.log
