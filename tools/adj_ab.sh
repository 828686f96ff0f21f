#!/bin/bash
# A/B of the signature adjacency lists (ANX_SCAN_ADJ=0 / 1): parity tests, then the bench line of both.  usage: adj_ab.sh <tag> ["pytest args"]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-adj}; PT=${2:-tests/test_gpu_parity.py -m gpu}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 1500 python -m pytest $PT -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest.log
for adj in 0 1 0 1; do
  ANX_ADJ_TIMING=1 ANX_SCAN_ADJ=$adj python3 $R/bench.py --no-extras --cpu-sample 0 --timed-only --steps 20 2>$O/bench_$adj.err | tail -1 > $O/bench_$adj.json
  python3 -c "
import json,sys
j=json.loads(open('$O/bench_$adj.json').readline()); print('adj=$adj', round(j['ms_per_step'],3), {k:round(v,3) for k,v in j['roofline']['kernels_ms'].items()}, j.get('pair_slots'), j.get('value'))" | tee -a $O/summary.txt
  grep "anx adjacency" $O/bench_$adj.err | tail -7 | tee -a $O/summary.txt
done
