#!/bin/bash
# One report with every number quoted in DESIGN.md (run on the GPU box):
#   bash tools/run_all_benchmarks.sh > gpurun_out/all_configs.txt
echo "== bench.py (config 4: 6-layer Llama3, fwd+bwd+Adam), per-GPU batch sweep"
for b in 64 128 256 512; do
  python bench.py --steps 5 --warmup 2 --batch $b --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(f\"batch {d['config']['per_gpu_batch']:4d}  {d['value']:8.1f} samples/s  {d['ms_per_step']:7.2f} ms/step  model {100*d['model_flops_frac_of_fp32_mfma_peak']:.1f} % of fp32-MFMA peak  GEMM roofline frac {d['roofline']['frac']:.3f}\")"
done
echo "== configs 2 / 3 and the GRU sequence (tools/bench_configs.py)"
python tools/bench_configs.py 10 2>/dev/null | grep -v amdgpu
echo "== greedy decode (tools/bench_decode.py)"
python tools/bench_decode.py 256 8 2>/dev/null | tail -1
echo "== per-shape GEMM table at the bench default (tools/gemm_shapes.py 256)"
python tools/gemm_shapes.py 256 2>/dev/null | grep -v amdgpu
echo "== fused attention kernels alone (B*H = 1536)"
bash tools/prof_cmd.sh attn python tools/attn_one.py 5 256 2>/dev/null | head -5
