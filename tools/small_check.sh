#!/bin/bash
# the small call: its tests, the call by batch size (python threads), native threads
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/small build
timeout 600 python -m pytest tests/test_gpu_small.py -q -x 2>&1 | tail -2
python3 tools/fresh_batch.py small 2>/dev/null | tail -1 | python3 -c "
import json,sys; j=json.loads(sys.stdin.read())['by_batch_size']; print({k:(round(v['best_us'],1) if 'best_us' in v else round(v['queries_per_s']/1e6,2)) for k,v in j.items()})"
g++ -O2 -std=c++17 -pthread -I include tools/small_threads.cpp -o build/small_threads -L analiticcl_amd -lanx -Wl,-rpath,$R/analiticcl_amd 2>/dev/null
D=/tmp/anx_bench_data_$(id -u)_0
python3 -c "
import sys; sys.path.insert(0,'$R')
from analiticcl_amd import synth
p=synth.materialize_golden('$D'); w=synth.load_lexicon_words(p['eng']); q=synth.make_queries(w,16000,max_len=16,seed=synth.SEED)
open('/tmp/q16k.txt','w').write('\n'.join(q)+'\n'); print(p['alphabet'], p['eng'])" > /tmp/paths.txt
read A L < /tmp/paths.txt
for t in 1 4 8 16; do ./build/small_threads $A $L /tmp/q16k.txt $t 1000 200 | tail -1; done
