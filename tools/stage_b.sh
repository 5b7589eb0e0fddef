#!/bin/bash
# round 4, stage B: persistent attention (forward + backward) -- tests, probe, same-box A/B of the bench
R=$PWD; O=$R/gpurun_out/stage_b; mkdir -p $O; rm -f $O/ab.txt
timeout 900 python -m pytest tests/test_fused_epilogues.py tests/test_kernels_gpu.py tests/test_llama_golden.py tests/test_batch_gate.py -m gpu -q -x 2>&1 | tail -15 > $O/tests.txt; cat $O/tests.txt
timeout 300 python tools/epilogue_probe.py 2>&1 | grep attention | tee $O/probe.txt
PDN_ATT_NO_PERSIST=1 timeout 300 python tools/epilogue_probe.py 2>&1 | grep attention | tee -a $O/probe.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', round(d['value'],1), round(d['ms_per_step'],3), d['final_loss'], d['batch_gate']['worst_grad_rel_err'] if d.get('batch_gate') else None, round(r['frac'],3), round(r['all_gemm']['frac'],3))"; }
for i in 1 2; do
  python bench.py --no-cpu-baseline 2>$O/err_new.txt | line new >> $O/ab.txt
  PDN_ATT_NO_PERSIST=1 python bench.py --no-cpu-baseline 2>/dev/null | line no_persist >> $O/ab.txt
done
cat $O/ab.txt; tail -5 $O/err_new.txt
bash tools/prof_cmd.sh r04b_bench python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gemm-prof --no-parity-gate > $O/bench_kernel_stats.txt 2>&1; head -30 $O/bench_kernel_stats.txt
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $O/tests_all.txt; cat $O/tests_all.txt
