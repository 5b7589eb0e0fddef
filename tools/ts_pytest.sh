#!/bin/bash
# `pytest -m gpu` with a wall-clock stamp per output line (gpurun_out/ts.log): where does a slow run of the suite spend its time?
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
nproc > gpurun_out/ts_host.txt; uptime >> gpurun_out/ts_host.txt
timeout 1400 python -u -m pytest tests -m gpu -v -p no:cacheprovider --durations=25 2>&1 | python -u -c "
import sys,time
t0=time.time(); last=t0
for l in sys.stdin:
    now=time.time()
    print('%7.1f %6.1f %s'%(now-t0, now-last, l.rstrip()[:170])); last=now
" > gpurun_out/ts.log
uptime >> gpurun_out/ts_host.txt
tail -1 gpurun_out/ts.log; sort -k2 -rn gpurun_out/ts.log | head -8; cat gpurun_out/ts_host.txt
