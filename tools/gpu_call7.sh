#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/call7
mkdir -p $O
cd $R
timeout 1700 python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest_all.log 2>&1; echo "pytest all rc=$?" | tee -a $O/summary.txt
tail -18 $O/pytest_all.log
timeout 400 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench.json; python - <<'PY'
import json
j=json.load(open("gpurun_out/call7/bench.json"))
print({k:j[k] for k in ("value","queries_per_s","ms_per_step","dp_pairs_per_s","e2e_queries_per_s","e2e","stage_ms","overlapped")})
print(j["roofline"]["kernel"], j["roofline"]["frac"], j["roofline"]["per_kernel"])
PY
tail -3 $O/bench.log | head -2
