#!/bin/bash
# k_filter_score by parts: a build with the debug switches (tools/build_flags.sh dbg "-DANX_DEBUG_SWITCHES"), parts skipped through ANX_SCORE_DBG
R=${GRAFT_REPO_ROOT:-$(pwd)}
export ANX_LIB=$R/build/libanx_dbg.so
for v in ${FS_DBGS:-0 1 2 6 14}; do
  ANX_SCORE_DBG=$v python3 $R/bench.py --no-extras --cpu-sample 0 --timed-only --no-overlap --steps 10 2>/dev/null | tail -1 | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); print('dbg=$v', round(b['ms_per_step'],3), {k:round(x,3) for k,x in b['roofline']['kernels_ms'].items()})"
done
