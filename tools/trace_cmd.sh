#!/bin/bash
# per-kernel timing (rocprofv3 kernel trace) of an arbitrary python tool: trace_cmd.sh <name> <script.py> [args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/meas
N=$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace_$N
S=$1; shift
rocprofv3 --kernel-trace --stats -d $O/trace_$N -- python3 $R/$S "$@" > $O/trace_$N.log 2>&1
cd $R
find $O/trace_$N -name "*.db" | head -1 | xargs python3 profiles/summarize_rocpd.py > $O/trace_$N.md
head -24 $O/trace_$N.md
rm -rf $O/trace_$N
