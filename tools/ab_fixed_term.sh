#!/bin/bash
# same-box A/B of the residency-round term of the tile cost model (PDN_GEMM_NO_FIXED=1 = round-2 model)
cd "$(dirname "$0")/.."
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), d['unit'], round(d['ms_per_step'],3))"; }
for b in 64 128 256; do
  python bench.py --batch $b --no-cpu-baseline --no-batch-gate 2>/dev/null | line "llama b$b new"
  PDN_GEMM_NO_FIXED=1 python bench.py --batch $b --no-cpu-baseline --no-batch-gate 2>/dev/null | line "llama b$b old"
done
for c in "mlp --batch 8192" "mlp --batch 65536 --steps 20" "lenet --batch 4096 --steps 50"; do
  python bench.py --config $c --no-cpu-baseline 2>/dev/null | line "$c new"
  PDN_GEMM_NO_FIXED=1 python bench.py --config $c --no-cpu-baseline 2>/dev/null | line "$c old"
done
python tools/bench_llama_dims.py 512 8 1536 | head -1 | cut -c1-150
PDN_GEMM_NO_FIXED=1 python tools/bench_llama_dims.py 512 8 1536 | head -1 | cut -c1-150
