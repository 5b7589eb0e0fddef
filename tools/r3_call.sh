#!/bin/bash
# one gpurun call: GPU tests of the round's new paths + the per-config bench lines
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_llama_golden.py tests/test_graph_gpu.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r3_t2.log
for c in decode mlp lenet gru; do
  timeout 600 python bench.py --config $c --steps 200 --warmup 20 > gpurun_out/r3_bench_$c.json 2> gpurun_out/r3_bench_$c.err
  tail -c 400 gpurun_out/r3_bench_$c.err
done
timeout 600 python bench.py --config lenet --batch 4096 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r3_bench_lenet4096.json 2>> gpurun_out/r3_bench_lenet.err
timeout 300 python tools/bench_decode.py 256 8 > gpurun_out/r3_decode_tool.log 2>&1
cat gpurun_out/r3_t2.log gpurun_out/r3_decode_tool.log
