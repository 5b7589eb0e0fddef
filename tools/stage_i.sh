#!/bin/bash
# round 4, stage I: cross-entropy statistics split over the projection (row maxima) and the input-gradient product (sum of
# exponentials, run in the forward pass) -- tests, same-box A/B
R=$PWD; O=$R/gpurun_out/stage_i; mkdir -p $O; rm -f $O/ab.txt
timeout 1200 python -m pytest tests/test_linear_ce.py tests/test_fused_epilogues.py tests/test_batch_gate.py tests/test_llama_golden.py tests/test_fullsize_properties_gpu.py -m gpu -q -x 2>&1 | tail -6 | tee $O/tests.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
f=r['other_gemm_families']; f[r['kernel']]=r
print('$1', round(d['value'],1), round(d['ms_per_step'],3), d['final_loss'], d['batch_gate']['worst_grad_rel_err'], {k[5:-7]:(round(v['frac'],3), round(v['time_share_of_step'],3)) for k,v in f.items()})"; }
for i in 1 2; do
  python bench.py --no-cpu-baseline 2>$O/err_new.txt | line deferred >> $O/ab.txt
  PDN_NO_CE_DEFERRED=1 python bench.py --no-cpu-baseline 2>/dev/null | line lse_in_gemm >> $O/ab.txt
  PDN_NO_CE_DEFERRED=1 PDN_NO_LSE_EPILOGUE=1 python bench.py --no-cpu-baseline 2>/dev/null | line stats_pass >> $O/ab.txt
done
cat $O/ab.txt; tail -3 $O/err_new.txt
bash tools/prof_cmd.sh r04i_bench python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gemm-prof --no-parity-gate > $O/bench_kernel_stats.txt 2>&1; head -24 $O/bench_kernel_stats.txt
