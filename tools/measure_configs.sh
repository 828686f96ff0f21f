#!/bin/bash
# Profiles of the configurations that are not the headline one (run on the GPU box through gpurun): per-kernel times (rocprofv3
# kernel trace) and PMC passes (each in its own run, --kernel-trace only) of
#   conf   : BASELINE configs[2]  (tools/conf_probe.py: nld.aspell, 1 M queries len <= 24, d = 3, 10 confusable patterns)
#   big    : BASELINE configs[3], one GPU's share (tools/big_lexicon_bench.py: 1 M-entry lexicon, 1.25 M queries)
#   search : BASELINE configs[4], one GPU's share (tools/search_bench.py: 12.5 MB of running text, max_ngram 3, bigram LM)
# Outputs gpurun_out/meas/<name>_kernel_trace.md and <name>_pmc.md; tools/collect_profiles.py copies them into profiles/.
# usage: measure_configs.sh [conf] [big] [search]      (default: all three)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/meas
mkdir -p $O
export ANX_RUN_OVERLAP=0   # every kernel alone on the GPU: clean per-kernel durations
what=${*:-conf big search}
cd /tmp && export TMPDIR=/tmp
for name in $what; do
  case $name in
    conf) cmd="$R/tools/conf_probe.py 1000000";;
    big) cmd="$R/tools/big_lexicon_bench.py 1000000 1250000 nocheck";;
    search) cmd="$R/tools/search_bench.py 12.5";;
    *) echo "unknown workload $name"; continue;;
  esac
  rm -rf $O/trace_$name $O/pmc_$name
  rocprofv3 --kernel-trace --stats -d $O/trace_$name -- python3 $cmd > $O/trace_$name.log 2>&1
  find $O/trace_$name -name "*.db" | head -1 | xargs python3 $R/profiles/summarize_rocpd.py > $O/${name}_kernel_trace.md
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$name/pass$i -- python3 $cmd > $O/pmc_${name}_pass$i.log 2>&1
  done
  mkdir -p $O/pmc_${name}_flat
  rm -rf $O/pmc_${name}_flat/*
  n=0; for f in $(find $O/pmc_$name -name "*counter_collection.csv"); do n=$((n+1)); mkdir -p $O/pmc_${name}_flat/p$n; cp $f $O/pmc_${name}_flat/p$n/pmc_counter_collection.csv; done
  python3 $R/profiles/summarize_pmc.py $O/pmc_${name}_flat > $O/${name}_pmc.md
  rm -rf $O/pmc_$name $O/trace_$name $O/pmc_${name}_flat
  echo "== $name"; head -8 $O/${name}_kernel_trace.md
done
