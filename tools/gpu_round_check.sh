#!/bin/bash
# One GPU call of the usual round checks: (a subset of) the GPU tests, then optional extra commands.  usage: gpu_round_check.sh <tag> "<pytest args>" ["cmd" ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
PT=$1; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 1700 python -m pytest $PT -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest.log
for c in "$@"; do echo "== $c" | tee -a $O/summary.txt; timeout 600 bash -c "$c" 2>&1 | tail -8 | tee -a $O/summary.txt; done
