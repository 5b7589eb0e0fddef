#!/bin/bash
# Everything the round's numbers come from, in one GPU-box call (outputs under gpurun_out/$ROUND/, default r03):
#   tests -m gpu, smoke, bench.py (default, forced single-rank DP, batch 128 / 64, --config mlp / lenet / gru / decode),
#   rocprofv3 kernel stats of the bench command, PMC passes (SQ / LDS / L2 / FETCH / WRITE) of the same command
#   -- regenerated EVERY time (the summary records a hash of the kernel sources it was measured on; bench.py flags
#   `traffic_stale` when the sources have changed since) -- the LeNet profile, attention / GEMM probes.
ROUND=${ROUND:-r06}
R=$PWD; O=$R/gpurun_out/$ROUND; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
# counters first: bench.py reads the summary it finds under profiles/ (copied there right away on this box)
bash tools/pmc_cmd.sh ${ROUND}_bench kernel python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gemm-prof --no-parity-gate --no-other-configs > $O/bench_pmc.txt 2>&1
python tools/stamp_pmc.py gpurun_out/pmc_${ROUND}_bench/summary.json $O/pmc_bench_default.json 512 && cp $O/pmc_bench_default.json profiles/${ROUND}_pmc_bench_default.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.json; echo
cp bench_detail.json $O/bench_detail.json                     # the full record the compact line names
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2>> $O/bench_default.err   # the driver's own command line
PDN_BENCH_FORCE_DP=1 python bench.py --no-cpu-baseline --no-other-configs > $O/bench_force_dp.json 2>> $O/bench_default.err
python bench.py --no-cpu-baseline --no-other-configs --batch 256 > $O/bench_b256.json 2>> $O/bench_default.err
python bench.py --no-cpu-baseline --no-other-configs --batch 1024 --steps 5 > $O/bench_b1024.json 2>> $O/bench_default.err
python bench.py --no-cpu-baseline --no-other-configs --batch 128 > $O/bench_b128.json 2>> $O/bench_default.err
python bench.py --no-cpu-baseline --no-other-configs --batch 64 > $O/bench_b64.json 2>> $O/bench_default.err
for c in mlp lenet gru decode; do python bench.py --config $c --steps 200 --warmup 20 > $O/bench_$c.json 2>> $O/bench_default.err; done
python bench.py --config transformer --steps 100 --warmup 10 > $O/bench_transformer.json 2>> $O/bench_default.err
python bench.py --config transformer --no-graph --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_transformer_eager.json 2>> $O/bench_default.err
# (rocprofv3 crashes on hipGraph launches: the profile times the eager launches and skips the launch-floor graph)
PDN_BENCH_NO_GRAPH_PROBE=1 bash tools/prof_cmd.sh ${ROUND}_transformer python bench.py --config transformer --no-graph --steps 20 --warmup 3 --no-cpu-baseline > $O/transformer_kernel_stats.txt 2>&1
# (the LeNet line reads the counter summary of its own kernels: collected and stamped first)
bash tools/pmc_cmd.sh ${ROUND}_lenet conv python tools/bench_configs.py 3 lenet:4096 > $O/lenet_pmc.txt 2>&1
python tools/stamp_pmc.py gpurun_out/pmc_${ROUND}_lenet/summary.json $O/pmc_lenet_b4096.json && cp $O/pmc_lenet_b4096.json profiles/${ROUND}_pmc_lenet_b4096.json
python bench.py --config lenet --batch 4096 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_lenet_b4096.json 2>> $O/bench_default.err
python bench.py --config mlp --batch 65536 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_mlp_b65536.json 2>> $O/bench_default.err
python bench.py --config mlp --batch 8192 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_mlp_b8192.json 2>> $O/bench_default.err
bash tools/prof_cmd.sh ${ROUND}_mlp python bench.py --config mlp --batch 65536 --steps 10 --warmup 3 --no-cpu-baseline > $O/mlp_b65536_kernel_stats.txt 2>&1
python tools/mlp_dw_probe.py > $O/mlp_dw_probe.txt 2>&1
bash tools/prof_cmd.sh ${ROUND}_bench python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gemm-prof --no-parity-gate --no-other-configs > $O/bench_kernel_stats.txt 2>&1
cp gpurun_out/prof_${ROUND}_bench/p_kernel_stats.csv $O/bench_default_kernel_stats.csv 2>/dev/null
bash tools/prof_cmd.sh ${ROUND}_lenet python tools/bench_configs.py 10 lenet:4096 > $O/lenet_kernel_stats.txt 2>&1
bash tools/prof_cmd.sh ${ROUND}_decode python tools/bench_decode.py 256 8 > $O/decode_kernel_stats.txt 2>&1
bash tools/pmc_cmd.sh ${ROUND}_attn attention python tools/attn_compare.py 256 256 48 > $O/attn_pmc.txt 2>&1
{ python tools/bench_configs.py 10; python tools/bench_graph.py 30; python tools/bench_decode.py 256 8; python tools/bench_decode.py 900 8;
  for a in "256 256 48" "128 512 48" "64 1024 48" "192 256 64" "96 512 64"; do python tools/attn_compare.py $a; done;
  python tools/gemm_shapes.py 256; python tools/gemm_shapes.py 64; bash tools/ab_small_batch.sh; bash tools/ab_fixed_term.sh;
  python tools/gemm_generic.py; python tools/bench_llama_dims.py 512 8 1536; python tools/bench_llama_dims.py 768 12 2048 128;
  python tools/bench_llama_dims.py 384 6 1024; python tools/bench_llama_dims.py 288 6 768 128 512; python tools/decode_probe.py;
  python tools/dp_overhead_probe.py dp; python tools/dp_overhead_probe.py base;
  python bench.py --config decode --steps 900 --warmup 20 --no-cpu-baseline; } > $O/all_configs.txt 2>&1
# phase timestamps from inside the decode kernels / the attention forward (traced builds made by the same scripts here)
bash tools/decode_trace.sh > $O/decode_trace.txt 2>&1
python tools/conv_quad_probe.py > $O/conv_quad_probe.txt 2>&1
PDN_CONV_QUAD=0 python tools/conv_quad_probe.py > $O/conv_direct_probe.txt 2>&1
{ python tools/plain_llama_bench.py 64 5; python tools/plain_llama_bench.py 256 4; } > $O/plain_llama_bench.txt 2>&1
{ for s in "4096 3200 500" "5632 512 512" "5632 512 1536" "8192 784 1024" "65536 784 1024" "65536 512 1536" "65536 288 768"; do echo "== rows, in, out: $s"; SWEEP_SHOW=1 python tools/gemm_fc_sweep.py $s; done; } > $O/gemm_fc_sweep.txt 2>&1
bash tools/prof_cmd.sh ${ROUND}_ew python tools/ew_strided_probe.py > $O/ew_strided_probe.txt 2>&1
PDN_EW_NO_FASTDIV=1 bash tools/prof_cmd.sh ${ROUND}_ew0 python tools/ew_strided_probe.py > $O/ew_strided_probe_64bit_divide.txt 2>&1
bash tools/prof_cmd.sh ${ROUND}_plain python tools/plain_llama_bench.py 256 4 plain > $O/plain_llama_kernel_stats.txt 2>&1
{ python tools/attn_hd128_probe.py 256 2048 1; python tools/attn_hd128_probe.py 256 2048 0; python tools/attn_hd128_probe.py 64 4096 0; python tools/attn_hd128_probe.py 1024 512 1; } > $O/attn_hd128_probe.txt 2>&1
python tools/epilogue_probe.py > $O/epilogue_probe.txt 2>&1
python tools/lmhead_probe.py > $O/lmhead_probe.txt 2>&1
python tools/attn_masked_probe.py > $O/attn_masked_probe.txt 2>&1
python tools/rowtile_probe.py > $O/rowtile_probe.txt 2>&1
python tools/lmhead_gap_probe.py > $O/lmhead_gap_probe.txt 2>&1
python tools/outres_fixed_probe.py > $O/outres_fixed_probe.txt 2>&1
bash tools/step_gaps.sh > $O/step_gaps.txt 2>&1
ls -la $O
