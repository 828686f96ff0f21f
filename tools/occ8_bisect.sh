#!/bin/bash
# Runs tools/occ8_probe.py per (d, max_len) under a timeout on the occupancy-forced build and on the product build.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/occ8
mkdir -p $O
for lib in occ8 product; do
  for cfg in "1 16" "2 16" "3 16" "2 28" "3 28" "1 28" "4 20"; do
    set -- $cfg
    ANX_PROBE_LIB=$lib timeout 150 python3 $R/tools/occ8_probe.py $1 $2 > $O/probe_${lib}_d$1_l$2.log 2>&1
    echo "$lib d=$1 max_len=$2 rc=$? $(tail -2 $O/probe_${lib}_d$1_l$2.log | tr '\n' ' ')" | tee -a $O/summary.txt
  done
done
