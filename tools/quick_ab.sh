#!/bin/bash
# Bench line (ms per step, per-kernel ms) of the in-tree library and of experiment builds: quick_ab.sh [tree|name ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
for n in "$@"; do
  if [ "$n" = tree ]; then unset ANX_LIB; else export ANX_LIB=$R/build/libanx_$n.so; fi
  for rep in 1 2; do
    python3 $R/bench.py --no-extras --cpu-sample 0 --timed-only --steps 20 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$n', round(j['ms_per_step'],3), {k:round(v,3) for k,v in j['roofline']['kernels_ms'].items()}, j['pair_slots'])"
  done
done
