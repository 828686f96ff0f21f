#!/bin/bash
# Round measurement set (run on the GPU box through gpurun): default bench line with CPU baseline, rocprofv3 kernel
# trace, PMC passes (each in its own run, --kernel-trace only).  Outputs under gpurun_out/meas/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/meas
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_default.log 2>&1
tail -1 $O/bench_default.log > $O/bench_default.json
rocprofv3 --kernel-trace --stats -d $O/trace -- python3 $R/bench.py --steps 5 --warmup 2 --timed-only --no-overlap > $O/trace.log 2>&1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc/pass$i -- python3 $R/bench.py --steps 2 --warmup 1 --timed-only --no-overlap > $O/pmc_pass$i.log 2>&1
done
cd $R
find $O/trace -name "*.db" | head -1 | xargs python3 profiles/summarize_rocpd.py > $O/kernel_trace.md
mkdir -p $O/pmc_flat
n=0; for f in $(find $O/pmc -name "*counter_collection.csv"); do n=$((n+1)); mkdir -p $O/pmc_flat/p$n; cp $f $O/pmc_flat/p$n/pmc_counter_collection.csv; done
python3 profiles/summarize_pmc.py $O/pmc_flat > $O/pmc.md
rm -rf $O/pmc $O/trace
cat $O/bench_default.json; head -12 $O/kernel_trace.md
