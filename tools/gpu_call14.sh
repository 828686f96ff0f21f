#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/call14
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_encode.py tests/test_gpu_parity.py -x -q > $O/pytest_a.log 2>&1; echo "pytest a rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest_a.log
for v in "X=1" "ANX_SCAN_WALK=flat" "ANX_SIG_GROUPS=7" "ANX_SIG_GROUPS=8"; do
  echo "== $v" | tee -a $O/summary.txt
  env $v timeout 300 python tools/scan_probe.py 2>&1 | grep -E "^(default|scan dbg)" | tee -a $O/summary.txt
done
for v in "X=1" "ANX_SCAN_WALK=flat" "ANX_SIG_GROUPS=7" "ANX_SIG_GROUPS=8"; do
  env $v timeout 600 python tools/big_lexicon_bench.py > $O/big_$(echo $v | tr '=' '_').log 2>&1
  echo "big $v: $(grep -E 'encode|ms_scan|spot' $O/big_$(echo $v | tr '=' '_').log | tr '\n' ' ')" | tee -a $O/summary.txt
done
