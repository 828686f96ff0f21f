#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/call11
mkdir -p $O
cd $R
cp analiticcl_amd/libanx.so /tmp/libanx_product.so
for v in product loop product; do
  if [ "$v" = "product" ]; then cp /tmp/libanx_product.so analiticcl_amd/libanx.so; else cp build/libanx_$v.so analiticcl_amd/libanx.so; fi
  echo "== $v" | tee -a $O/summary.txt
  timeout 300 python tools/scan_probe.py 2>&1 | grep -E "^(default|score)" | tee -a $O/summary.txt
done
cp /tmp/libanx_product.so analiticcl_amd/libanx.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_config3.py -x -q > $O/pytest_a.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest_a.log
