#!/bin/bash
# Search-mode A/B on ONE box (call-level numbers differ by +-15 % between boxes): search_ab.sh <rounds> <name|tree> ...  (build/libanx_<name>.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}
rounds=$1; shift
for i in $(seq $rounds); do
  for n in "$@"; do
    if [ "$n" = tree ]; then unset ANX_LIB; else export ANX_LIB=$R/build/libanx_$n.so; fi
    echo "$n: $(timeout 200 python3 -u $R/tools/search_bench.py 12.5 4 2>&1 | tail -1)"
  done
done
