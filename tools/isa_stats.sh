#!/bin/bash
# Register / LDS / spill figures of the engine's kernels from the gfx950 ISA metadata (no GPU needed): isa_stats.sh [name filter]
R=$(cd "$(dirname "$0")/.." && pwd)
S=${ISA_OUT:-/tmp/engine_isa.s}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -o $S $R/analiticcl_amd/csrc/${ISA_SRC:-engine.hip} 2>/dev/null
python3 - "$S" "${1:-}" <<'PY'
import re, sys
text = open(sys.argv[1]).read()
meta = text[text.index("amdhsa.kernels:"):]
for blk in re.split(r"\n  - ", meta)[1:]:
    nm = re.search(r"\.name:\s+(\S+)", blk)
    if not nm or ".vgpr_count" not in blk: continue
    name = nm.group(1)
    if sys.argv[2] and sys.argv[2] not in name: continue
    g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
    print(f"{name[:64]:64s} vgpr {g('vgpr_count'):3d} sgpr {g('sgpr_count'):3d} lds {g('group_segment_fixed_size'):6d} spill v{g('vgpr_spill_count')} s{g('sgpr_spill_count')} scratch {g('private_segment_fixed_size')}")
PY
