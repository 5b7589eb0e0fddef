#!/bin/bash
# idle time between consecutive kernels of one training step (rocprofv3 kernel trace of bench.py): bash tools/step_gaps.sh
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ht; rocprofv3 --kernel-trace --output-format csv -d /tmp/ht -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gemm-prof --no-parity-gate --no-other-configs "$@" > /tmp/ht.log 2>&1
f=$(find /tmp/ht -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv, collections
rows=list(csv.DictReader(open("$f")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"] for r in rows]
idx=[i for i,n in enumerate(names) if n.startswith("adam_multi")]
a,b=idx[-2],idx[-1]
S=[int(r["Start_Timestamp"]) for r in rows]; E=[int(r["End_Timestamp"]) for r in rows]
span=(E[b]-E[a])/1e3; busy=sum(E[i]-S[i] for i in range(a+1,b+1))/1e3
gaps=[(S[i]-E[i-1])/1e3 for i in range(a+1,b+1)]
print(f"one step: {b-a} kernels, span {span:.1f} us, kernel time {busy:.1f} us, idle between kernels {sum(g for g in gaps if g>0):.1f} us (overlap {-sum(g for g in gaps if g<0):.1f} us)")
big=sorted(((g,i) for g,i in zip(gaps,range(a+1,b+1))),reverse=True)[:12]
for g,i in big: print(f"  {g:7.1f} us before {names[i][:70]}  (after {names[i-1][:50]})")
h=collections.Counter(min(int(g),10) for g in gaps if g>0)
print("gap histogram (us, 10 = 10 or more):", sorted(h.items()))
PY
