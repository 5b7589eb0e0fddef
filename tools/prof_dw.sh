#!/bin/bash
# kernel-trace of the weight-gradient GEMMs only (tools/gemm_sweep_dw.py --auto): per-kernel durations
R=$PWD; OUT=$R/gpurun_out/prof_dw; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o dw -- python $R/tools/gemm_sweep_dw.py 32768 auto > $OUT/log.txt 2>&1
cd $R
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cut -d, -f1-4 $f | cut -c1-150 | head -12
