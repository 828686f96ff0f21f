mkdir -p gpurun_out/meas
for v in 0 1 2; do
  ANX_SCORE_DBG=$v python bench.py --cpu-sample 0 --steps 5 2>/dev/null | tail -1 | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print('dbg=$v', b['ms_per_step'], b['roofline']['kernels_ms'], b['stage_ms'])"
done
ANX_PREFILTER=0 python bench.py --cpu-sample 0 --steps 5 2>/dev/null | tail -1 | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print('prefilter=0', b['ms_per_step'], b['roofline']['kernels_ms'], b['stage_ms'], b.get('dl_pairs'))"
