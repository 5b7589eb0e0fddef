#!/bin/bash
# round 4, stage K: short per-GPU batches after this round's kernel changes
R=$PWD; O=$R/gpurun_out/stage_k; mkdir -p $O
for b in 64 128; do
  python bench.py --batch $b --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_b$b.json
  python -c "
import json; d=json.load(open('$O/bench_b$b.json')); r=d['roofline']; f=r['other_gemm_families']; f[r['kernel']]=r
print('B=$b', round(d['value'],1), round(d['ms_per_step'],3), {k[5:-7]:(round(v['frac'],3), round(v['time_share_of_step'],3)) for k,v in f.items()})"
  bash tools/prof_cmd.sh r04k_b$b python bench.py --batch $b --steps 3 --warmup 1 --no-cpu-baseline --no-gemm-prof --no-parity-gate > $O/kernel_stats_b$b.txt 2>&1; head -30 $O/kernel_stats_b$b.txt
done
