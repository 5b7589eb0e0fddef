#!/bin/bash
# round 4, stage L: split cross-entropy statistics at short batches (vocabulary ranges over the grid) -- tests, A/B
R=$PWD; O=$R/gpurun_out/stage_l; mkdir -p $O; rm -f $O/ab.txt
timeout 1200 python -m pytest tests/test_linear_ce.py tests/test_fused_epilogues.py tests/test_batch_gate.py tests/test_llama_golden.py -m gpu -q -x 2>&1 | tail -6 | tee $O/tests.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
f=r['other_gemm_families']; f[r['kernel']]=r
print('$1', round(d['value'],1), round(d['ms_per_step'],3), d['final_loss'], d['batch_gate']['worst_grad_rel_err'], {k[5:-7]:(round(v['frac'],3), round(v['time_share_of_step'],3)) for k,v in f.items()})"; }
for b in 64 128 256; do
  python bench.py --batch $b --no-cpu-baseline 2>$O/err_new.txt | line "B=$b deferred" >> $O/ab.txt
  PDN_NO_CE_DEFERRED=1 PDN_NO_LSE_EPILOGUE=1 python bench.py --batch $b --no-cpu-baseline 2>/dev/null | line "B=$b stats_pass" >> $O/ab.txt
done
cat $O/ab.txt; tail -3 $O/err_new.txt
