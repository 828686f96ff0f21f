#!/bin/bash
# Fresh-batch measurement set (gpurun): modes of tools/fresh_batch.py untraced, then kernel traces of loop / encode / rerun.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-fresh}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in loop encode run rerun small; do
  timeout 600 python3 $R/tools/fresh_batch.py $m 20 > $O/$m.json 2> $O/$m.err
  tail -1 $O/$m.json
done
for m in loop encode rerun run; do
  timeout 600 rocprofv3 --kernel-trace -d $O/trace_$m -- python3 $R/tools/fresh_batch.py $m 12 > $O/trace_$m.log 2>&1
done
cd $R
L=$(find $O/trace_loop -name "*.db" | head -1); E=$(find $O/trace_encode -name "*.db" | head -1); RR=$(find $O/trace_rerun -name "*.db" | head -1); RU=$(find $O/trace_run -name "*.db" | head -1)
python3 tools/timeline.py $L $RU $E $RR > $O/timeline_loop.md
python3 tools/timeline.py $E > $O/timeline_encode.md
python3 tools/timeline.py $RU > $O/timeline_run.md
python3 tools/timeline.py $RR > $O/timeline_rerun.md
# keep the loop database (small) for a closer look; drop the rest
cp $L $O/loop.db 2>/dev/null
rm -rf $O/trace_loop $O/trace_encode $O/trace_rerun $O/trace_run
head -40 $O/timeline_loop.md
