#!/bin/bash
# tests -m gpu (summary line) + two default bench runs, one line each -> gpurun_out/quick.txt
R=$PWD; O=$R/gpurun_out/quick.txt; : > $O
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -5 >> $O
for i in 1 2; do
  python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(round(d['value'],1), round(d['ms_per_step'],3), d['final_loss'], d['parity_gate']['rel_err'] if d.get('parity_gate') else None, r['kernel'], round(r['frac'],3), round(r['all_gemm']['frac'],3), {k:(round(v['frac'],3), round(v['time_share_of_step'],3)) for k,v in r['other_gemm_families'].items()})" >> $O
done
cat $O
