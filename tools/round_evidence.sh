#!/bin/bash
# Everything the round's numbers come from, in one GPU-box call (outputs under gpurun_out/r02/):
#   tests -m gpu, smoke, bench.py (default, forced single-rank DP, batch 128 / 64), rocprofv3 kernel stats of the
#   bench command, PMC passes (SQ / LDS / L2 / FETCH / WRITE) of the same command, the other BASELINE configs.
R=$PWD; O=$R/gpurun_out/r02; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json; echo
PDN_BENCH_FORCE_DP=1 python bench.py --no-cpu-baseline > $O/bench_force_dp.json 2>> $O/bench_default.err
python bench.py --no-cpu-baseline --batch 128 > $O/bench_b128.json 2>> $O/bench_default.err
python bench.py --no-cpu-baseline --batch 64 > $O/bench_b64.json 2>> $O/bench_default.err
bash tools/prof_cmd.sh r02_bench python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gemm-prof --no-parity-gate > $O/bench_kernel_stats.txt 2>&1
cp gpurun_out/prof_r02_bench/p_kernel_stats.csv $O/bench_b256_kernel_stats.csv 2>/dev/null
bash tools/pmc_cmd.sh r02_bench kernel python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gemm-prof --no-parity-gate > $O/bench_pmc.txt 2>&1
cp gpurun_out/pmc_r02_bench/summary.json $O/pmc_bench_b256.json 2>/dev/null
bash tools/prof_cmd.sh r02_lenet python tools/bench_configs.py 10 lenet:4096 > $O/lenet_kernel_stats.txt 2>&1
bash tools/pmc_cmd.sh r02_lenet conv python tools/bench_configs.py 3 lenet:4096 > $O/lenet_pmc.txt 2>&1
python tools/bench_configs.py 10 > $O/configs.txt 2>&1
python tools/bench_graph.py 30 > $O/graph.txt 2>&1
python tools/bench_decode.py > $O/decode.txt 2>&1
python tools/gemm_shapes.py 256 > $O/gemm_shapes_b256.txt 2>&1
python tools/attn_compare.py 256 > $O/attn_compare.txt 2>&1
python tools/dp_overhead_probe.py dp > $O/dp_probe.txt 2>&1; python tools/dp_overhead_probe.py base >> $O/dp_probe.txt 2>&1
python tools/two_stream_probe.py 65536 > $O/two_stream.txt 2>&1
NO_EPI=1 python tools/rowres_probe.py 65536 > $O/rowres_probe.txt 2>&1
python tools/outres_probe.py 65536 > $O/outres_probe.txt 2>&1
tools/micro/mfma_sustained.bin > $O/mfma_sustained.txt 2>&1
ls -la $O
