#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/call13
mkdir -p $O
cd $R
bash tools/measure_round.sh > $O/measure.log 2>&1; tail -14 $O/measure.log | cut -c1-400
