#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/call6
mkdir -p $O
cd $R
ANX_ENCODE_TIMING=1 timeout 300 python tools/e2e_timing.py > $O/e2e.log 2>&1; grep -E "anx encode|from a" $O/e2e.log | tail -24 | tee -a $O/summary.txt
timeout 1700 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1; echo "pytest all rc=$?" | tee -a $O/summary.txt
tail -8 $O/pytest_all.log
timeout 400 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench.json; python - <<'PY'
import json
j=json.load(open("gpurun_out/call6/bench.json"))
print({k:j[k] for k in ("value","queries_per_s","ms_per_step","dp_pairs_per_s","e2e_queries_per_s","e2e","stage_ms")})
print(j["roofline"]["kernel"], j["roofline"]["frac"], j["roofline"]["per_kernel"])
PY
