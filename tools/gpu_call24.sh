#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/call24
mkdir -p $O
cd $R
timeout 1700 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1; echo "pytest all rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest_all.log
bash tools/measure_round.sh > $O/measure.log 2>&1; tail -8 $O/measure.log | cut -c1-300
