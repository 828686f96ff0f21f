#!/bin/bash
# quick per-kernel timing of the bench (rocprofv3 kernel trace); extra env via "$@" as VAR=VALUE words
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/meas
mkdir -p $O
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace_q
rocprofv3 --kernel-trace --stats -d $O/trace_q -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-sample 0 > $O/trace_q.log 2>&1
cd $R
find $O/trace_q -name "*.db" | head -1 | xargs python3 profiles/summarize_rocpd.py | head -16
rm -rf $O/trace_q
