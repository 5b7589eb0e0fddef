#!/bin/bash
# Run on the GPU box:  bash tools/profile_bench.sh <tag> [batch]
# 1) rocprofv3 --kernel-trace --stats of a short bench.py run  -> gpurun_out/prof_<tag>/stats
# 2) two separate PMC passes (FETCH_SIZE, WRITE_SIZE cannot share a pass on gfx950) with
#    kernel-trace only, aggregated per kernel by tools/summarize_pmc.py
# Summaries worth keeping are copied into profiles/ by hand (gpurun_out/ is scratch).
set -u
TAG=${1:-r01}
B=${2:-256}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 3 --warmup 1 --batch $B --no-cpu-baseline --no-gemm-prof"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $CMD > $OUT/stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -o bench -- $CMD > $OUT/pmc_$C.log 2>&1
done
cd $R
find $OUT -name "*.csv" | head -20
python tools/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
tail -40 $OUT/summary.txt
