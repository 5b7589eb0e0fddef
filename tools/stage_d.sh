#!/bin/bash
# round 4, stage D: A/B of two library builds (PDN_LIB) on the attention probe and the bench
R=$PWD; O=$R/gpurun_out/stage_d; mkdir -p $O; rm -f $O/ab.txt
timeout 600 python -m pytest tests/test_fused_epilogues.py tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | tail -4 | tee $O/tests.txt
for i in 1 2; do
  echo new; timeout 300 python tools/epilogue_probe.py 2>&1 | grep "rotated" | tee -a $O/probe.txt
  echo prev; PDN_LIB=$R/pydynet_amd/libpdnhip_prev.so timeout 300 python tools/epilogue_probe.py 2>&1 | grep "rotated" | tee -a $O/probe.txt
done
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), round(d['ms_per_step'],3), d['final_loss'])"; }
for i in 1 2; do
  python bench.py --no-cpu-baseline --no-gemm-prof 2>$O/err_new.txt | line new >> $O/ab.txt
  PDN_LIB=$R/pydynet_amd/libpdnhip_prev.so python bench.py --no-cpu-baseline --no-gemm-prof 2>/dev/null | line prev >> $O/ab.txt
done
cat $O/ab.txt; tail -3 $O/err_new.txt
python bench.py --config lenet --batch 4096 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_lenet_b4096.json 2>$O/err_lenet.txt; tail -c 1500 $O/bench_lenet_b4096.json; tail -3 $O/err_lenet.txt
