#!/bin/bash
# round 4, stage C: lm_head forward NT + quad-interleaved X image in the weight-gradient kernel -- tests, same-box A/B
R=$PWD; O=$R/gpurun_out/stage_c; mkdir -p $O; rm -f $O/ab.txt
timeout 900 python -m pytest tests/test_fused_epilogues.py tests/test_kernels_gpu.py tests/test_linear_ce.py tests/test_llama_golden.py tests/test_batch_gate.py -m gpu -q -x 2>&1 | tail -15 > $O/tests.txt; cat $O/tests.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
f=r['other_gemm_families']; f[r['kernel']]=r
print('$1', round(d['value'],1), round(d['ms_per_step'],3), d['final_loss'], d['batch_gate']['worst_grad_rel_err'] if d.get('batch_gate') else None, {k[5:-7]:(round(v['frac'],3), round(v['time_share_of_step'],3)) for k,v in f.items()}, 'fused', round(r['fused_epilogue_gemms']['frac'],3), round(r['fused_epilogue_gemms']['hbm_frac'],3))"; }
for i in 1 2; do
  python bench.py --no-cpu-baseline 2>$O/err_new.txt | line new >> $O/ab.txt
  PDN_OUTRES_TN_XQ=0 python bench.py --no-cpu-baseline 2>/dev/null | line no_xq >> $O/ab.txt
  PDN_LINCE_NT=0 python bench.py --no-cpu-baseline 2>/dev/null | line no_nt >> $O/ab.txt
done
cat $O/ab.txt; tail -5 $O/err_new.txt
python tools/gemm_shapes.py 256 > $O/gemm_shapes_256.txt 2>&1; PDN_OUTRES_TN_XQ=0 python tools/gemm_shapes.py 256 > $O/gemm_shapes_256_noxq.txt 2>&1; cat $O/gemm_shapes_256.txt; grep -i "dW\|tn" $O/gemm_shapes_256_noxq.txt
