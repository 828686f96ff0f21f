#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/call1
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
timeout 600 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?" | tee -a $O/summary.txt; tail -1 $O/bench.log > $O/bench.json
ANX_FS_SPLIT=0 timeout 300 python bench.py --cpu-sample 0 > $O/bench_nosplit.log 2>&1; tail -1 $O/bench_nosplit.log > $O/bench_nosplit.json
bash tools/occ8_bisect.sh
cp analiticcl_amd/libanx.so /tmp/libanx_product.so
cp build/libanx_occ8.so analiticcl_amd/libanx.so
ANX_FS_SPLIT=0 timeout 1200 python -m pytest tests -m gpu -q --timeout 150 --timeout-method thread -p no:cacheprovider > $O/pytest_occ8.log 2>&1; echo "pytest occ8 rc=$?" | tee -a $O/summary.txt
cp /tmp/libanx_product.so analiticcl_amd/libanx.so
tail -5 $O/pytest_gpu.log; tail -15 $O/pytest_occ8.log; cat $R/gpurun_out/occ8/summary.txt
python - <<'PY'
import json
for f in ("bench.json","bench_nosplit.json"):
    try:
        j=json.load(open("gpurun_out/call1/"+f)); print(f, j["value"], j["ms_per_step"], j["stage_ms"], j["roofline"]["kernels_ms"])
    except Exception as e: print(f, "ERR", e)
PY
