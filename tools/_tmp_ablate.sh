for a in 0 1 2 3; do PDN_QUAD_ABLATE=$a python bench.py --config lenet --batch 4096 --steps 30 --warmup 5 --no-cpu-baseline --no-parity-gate 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ablate $a', d['ms_per_step'], d['roofline']['conv_kernels']['conv2_bwd_data']['avg_launch_us'])"; done
