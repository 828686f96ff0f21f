#!/bin/bash
# A/B: the inline 8-word prefilter (k_filter_score<D, true>) against k_filter_wide for batches with long queries (ANX_FS_SPLIT=2)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for s in 1 2; do
  echo "== ANX_FS_SPLIT=$s configs[2]"; ANX_FS_SPLIT=$s python3 $R/tools/conf_probe.py 1000000 2>&1 | tail -4
  echo "== ANX_FS_SPLIT=$s configs[3] share"; ANX_FS_SPLIT=$s python3 $R/tools/big_lexicon_bench.py 1000000 1250000 nocheck 2>&1 | tail -3
done
