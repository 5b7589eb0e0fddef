#!/bin/bash
# round 4, stage E: lean conv weight gradient -- tests, LeNet bench A/B, kernel stats
R=$PWD; O=$R/gpurun_out/stage_e; mkdir -p $O; rm -f $O/ab.txt
timeout 900 python -m pytest tests/test_conv_direct_gpu.py tests/test_conv_relu_pool.py tests/test_frontend_parity.py tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | tail -12 | tee $O/tests.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['conv_kernels']
print('$1', round(d['value']), round(d['ms_per_step'],3), {n:(round(v['avg_launch_us'],1), round(v['mfma_frac'],3)) for n,v in k.items()})"; }
for i in 1 2; do
  python bench.py --config lenet --batch 4096 --steps 50 --warmup 5 --no-cpu-baseline 2>$O/err.txt | line lean >> $O/ab.txt
  PDN_CONV_WGRAD_LEAN=0 python bench.py --config lenet --batch 4096 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | line old >> $O/ab.txt
done
cat $O/ab.txt; tail -3 $O/err.txt
bash tools/prof_cmd.sh r04e_lenet python tools/bench_configs.py 10 lenet:4096 > $O/lenet_kernel_stats.txt 2>&1; head -14 $O/lenet_kernel_stats.txt
