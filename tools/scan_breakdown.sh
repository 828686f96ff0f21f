#!/bin/bash
# k_scan_bits time with parts of the kernel skipped (ANX_SCAN_DBG bits: 1 one query per pass, 2 no class tests,
# 4 signature scan only, 8 no hit expansion); results are wrong when set -- timing only
for v in 0 8 1 2 4; do
  ANX_SCAN_DBG=$v python bench.py --cpu-sample 0 --steps 5 2>/dev/null | tail -1 | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print('scan dbg=$v', b['roofline']['kernels_ms']['k_scan_bits'])"
done
