#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/call9
mkdir -p $O
cd $R
bash tools/measure_round.sh > $O/measure.log 2>&1; tail -16 $O/measure.log
cd $R
timeout 600 python tools/big_lexicon_bench.py > $O/big.log 2>&1; grep -E "encode|ms_scan|spot|lexicon|build" $O/big.log | tee -a $O/summary.txt
timeout 600 python tools/search_bench.py 12.5 > $O/search.log 2>&1; tail -4 $O/search.log | tee -a $O/summary.txt
timeout 300 python tools/e2e_timing.py nld confusables > $O/e2e_nld.log 2>&1; tail -4 $O/e2e_nld.log | tee -a $O/summary.txt
