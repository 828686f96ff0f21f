#!/bin/bash
# The round's whole measurement set in one GPU call: measure_round.sh (bench line, kernel trace, PMC passes of configs[1]),
# measure_configs.sh (configs[2]-[4]), fresh_measure.sh (the fresh-batch step taken apart), the small call's kernels in stream order.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/measure_round.sh > gpurun_out/final_round.log 2>&1
bash tools/measure_configs.sh > gpurun_out/final_configs.log 2>&1
bash tools/fresh_measure.sh fresh_final > gpurun_out/final_fresh.log 2>&1
mkdir -p gpurun_out/small_final
cd /tmp && export TMPDIR=/tmp
for n in 1 64 1000; do
  rm -rf $R/gpurun_out/small_final/t$n
  rocprofv3 --kernel-trace -d $R/gpurun_out/small_final/t$n -- python3 $R/tools/small_trace.py $n 20 > $R/gpurun_out/small_final/trace_$n.log 2>&1
  db=$(find $R/gpurun_out/small_final/t$n -name "*.db" | head -1)
  (cd $R && python3 tools/dump_last_call.py $db > gpurun_out/small_final/call_$n.txt 2>&1)
  rm -rf $R/gpurun_out/small_final/t$n
done
cd $R
bash tools/small_check.sh > gpurun_out/small_final/check.txt 2>&1
tail -3 gpurun_out/final_round.log; tail -3 gpurun_out/small_final/check.txt
