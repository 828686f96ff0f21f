#!/bin/bash
# The 3-ranks-on-one-GPU bench job repeated (an intermittent hang was seen once in eight runs): dist_loop.sh [runs]; a hung run dumps every rank's Python stacks
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/dist
for i in $(seq 1 ${1:-12}); do
  s=$(date +%s)
  ANX_BENCH_WATCHDOG=90 MASTER_PORT=$((29600 + i)) timeout 200 python3 bench.py --ranks-on-one-gpu 3 --backend gloo --check-gather --queries 120000 --steps 3 --warmup 1 --cpu-sample 2000 > gpurun_out/dist/out_$i.log 2> gpurun_out/dist/err_$i.log; rc=$?
  echo "run $i rc=$rc $(( $(date +%s) - s )) s"
  if [ $rc -ne 0 ]; then grep -n "File \|Thread\|line " gpurun_out/dist/err_$i.log | tail -60; fi
done
