#!/bin/bash
# Stage times (every kernel alone on the GPU, HIP events) of the in-tree library and of experiment builds: stage_ab.sh [tree|name ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
for n in "$@"; do
  if [ "$n" = tree ]; then unset ANX_LIB; else export ANX_LIB=$R/build/libanx_$n.so; fi
  for rep in 1 2; do
    python3 $R/tools/fresh_batch.py kernels 30 2>/dev/null | tail -1 | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$n', {k:round(v,3) for k,v in j['stage_ms'].items()})"
  done
done
