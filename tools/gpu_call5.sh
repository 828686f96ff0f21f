#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/call5
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_encode.py -x -q > $O/pytest_enc.log 2>&1; echo "pytest encode rc=$?" | tee -a $O/summary.txt
tail -15 $O/pytest_enc.log
ANX_ENCODE_TIMING=1 timeout 300 python tools/e2e_timing.py > $O/e2e.log 2>&1; tail -4 $O/e2e.log | tee -a $O/summary.txt
ANX_ENCODE=host timeout 300 python tools/e2e_timing.py 2>&1 | tail -2 | tee -a $O/summary.txt
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_encode.py > $O/pytest_all.log 2>&1; echo "pytest all rc=$?" | tee -a $O/summary.txt
tail -8 $O/pytest_all.log
timeout 400 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench.json; python - <<'PY'
import json
j=json.load(open("gpurun_out/call5/bench.json"))
print({k:j[k] for k in ("value","queries_per_s","ms_per_step","dp_pairs_per_s","e2e_queries_per_s","e2e","stage_ms")})
print(j["roofline"]["kernel"], j["roofline"]["frac"], j["roofline"]["per_kernel"])
print(j["cpu_baseline"])
PY
