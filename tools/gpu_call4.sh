#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/call4
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_config2.py tests/test_gpu_config3.py tests/test_gpu_config4.py -x -q > $O/pytest_a.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -8 $O/pytest_a.log
for v in "X=1" "ANX_SIG_GROUPS=7" "ANX_SCAN_CHUNK=128" "ANX_SIG_GROUPS=7 ANX_SCAN_CHUNK=128" "ANX_SIG_GROUPS=7 ANX_SCAN_CHUNK=64" "ANX_SIG_GROUPS=8 ANX_SCAN_CHUNK=128"; do
  echo "== $v" | tee -a $O/summary.txt
  env $v timeout 300 python tools/scan_probe.py 2>&1 | grep -E "^(default|scan dbg)" | tee -a $O/summary.txt
done
