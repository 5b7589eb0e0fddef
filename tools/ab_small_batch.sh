#!/bin/bash
# same-box A/B of the small-batch switches (per-GPU batch 64 / 128): K split of the output-resident kernel, fused
# lm_head + cross-entropy node threshold
cd "$(dirname "$0")/.."
run() { python bench.py --batch $1 --no-cpu-baseline --no-batch-gate --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$2', d['config']['per_gpu_batch'], round(d['value']), round(d['ms_per_step'],2), round(d['model_flops_frac_of_fp32_mfma_peak'],4), 'all_gemm', round(r['all_gemm']['frac'],3))"; }
for b in 64 128; do
  for rep in 1 2; do
    run $b "split+ce16k      "
    PDN_OUTRES_NO_SPLIT=1 run $b "nosplit+ce16k    "
    PDN_LINCE_MIN_ROWS=32768 run $b "split+ce32k      "
    PDN_LINCE_MIN_ROWS=32768 PDN_OUTRES_NO_SPLIT=1 run $b "nosplit+ce32k(r2)"
  done
done
