#!/bin/bash
# kernel-trace + stats of an arbitrary command:  bash tools/prof_cmd.sh <tag> <command...>
TAG=$1; shift
R=$PWD; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- "$@" > $OUT/log.txt 2>&1 )
cd $R
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.3f} ms")
for r in rows[:22]:
    print(f'{r["Name"][:95]:95s} calls={r["Calls"]:>5s} total_ms={float(r["TotalDurationNs"])/1e6:9.3f} avg_us={float(r["AverageNs"])/1e3:9.1f} {r["Percentage"]:>6s}%')
PY
