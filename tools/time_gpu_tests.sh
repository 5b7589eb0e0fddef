#!/bin/bash
# wall time of the GPU suite file by file (where does a slow `pytest -m gpu` spend its time?)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/gpu_test_times.txt
: > $out
for f in tests/test_*.py; do
  s=$(date +%s%N)
  timeout 600 python -m pytest "$f" -m gpu -q -x -p no:cacheprovider > /tmp/t.log 2>&1
  rc=$?
  e=$(date +%s%N)
  printf "%8.1f s rc=%d %s  %s\n" "$(awk -v a=$s -v b=$e 'BEGIN { print (b - a) / 1e9 }')" $rc "$f" "$(tail -1 /tmp/t.log)" >> $out
done
sort -rn $out | head -40
