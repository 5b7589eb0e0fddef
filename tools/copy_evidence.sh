#!/bin/bash
# gpurun_out/$ROUND (scratch, written by tools/round_evidence.sh on the GPU box) -> profiles/ (tracked)
ROUND=${ROUND:-r06}; S=gpurun_out/$ROUND; D=profiles
cp $S/bench_default.json $D/${ROUND}_bench_default.json
cp $S/bench_detail.json $D/${ROUND}_bench_detail.json
cp $S/bench_driver_cmd.json $D/${ROUND}_bench_driver_cmd.json
for f in conv_quad_probe conv_direct_probe plain_llama_bench gemm_fc_sweep attn_hd128_probe transformer_kernel_stats ew_strided_probe ew_strided_probe_64bit_divide plain_llama_kernel_stats; do [ -f $S/$f.txt ] && cp $S/$f.txt $D/${ROUND}_$f.txt; done
tail -1 $S/bench_force_dp.json > $D/${ROUND}_bench_force_dp.json
for b in 1024 256 128 64; do cp $S/bench_b$b.json $D/${ROUND}_bench_b$b.json; done
for c in mlp lenet gru decode lenet_b4096 mlp_b65536 mlp_b8192 transformer transformer_eager; do cp $S/bench_$c.json $D/${ROUND}_bench_$c.json; done
cp $S/bench_default_kernel_stats.csv $D/${ROUND}_bench_default_kernel_stats.csv
cp $S/bench_kernel_stats.txt $D/${ROUND}_bench_default_kernel_stats.txt
cp $S/pmc_bench_default.json $D/${ROUND}_pmc_bench_default.json
cp $S/bench_pmc.txt $D/${ROUND}_pmc_bench_default.txt
cp $S/lenet_kernel_stats.txt $D/${ROUND}_lenet_b4096_kernel_stats.txt
cp $S/lenet_pmc.txt $D/${ROUND}_lenet_b4096_pmc.txt
cp $S/decode_kernel_stats.txt $D/${ROUND}_decode_kernel_stats.txt
cp $S/attn_pmc.txt $D/${ROUND}_attention_pmc.txt
cp $S/all_configs.txt $D/${ROUND}_all_configs.txt
cp $S/decode_trace.txt $D/${ROUND}_decode_trace.txt
cp $S/epilogue_probe.txt $D/${ROUND}_epilogue_probe.txt
cp $S/lmhead_probe.txt $D/${ROUND}_lmhead_probe.txt
cp $S/attn_masked_probe.txt $D/${ROUND}_attn_masked_probe.txt
for f in rowtile_probe lmhead_gap_probe outres_fixed_probe; do cp $S/$f.txt $D/${ROUND}_$f.txt; done
cp $S/step_gaps.txt $D/${ROUND}_step_gaps.txt
cp $S/mlp_b65536_kernel_stats.txt $D/${ROUND}_mlp_b65536_kernel_stats.txt
cp $S/mlp_dw_probe.txt $D/${ROUND}_mlp_dw_probe.txt
cp $S/pmc_lenet_b4096.json $D/${ROUND}_pmc_lenet_b4096.json
{ grep -E "passed|failed|error" $S/pytest_gpu.log | tail -2; tail -1 $S/smoke.log; } > $D/${ROUND}_gpu_tests.txt
ls -la $D | grep ${ROUND}_
