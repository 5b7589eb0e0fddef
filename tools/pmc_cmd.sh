#!/bin/bash
# PMC passes (SQ busy/wait/MFMA, LDS, L2 hit, FETCH/WRITE) over an arbitrary command; per-kernel averages.
#   bash tools/pmc_cmd.sh <tag> <kernel-name-substring> <command...>
# Counters are collected in their own runs with --kernel-trace only (no other trace domains).
TAG=$1; FILTER=$2; shift; shift
R=$PWD; OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"
P3="TCC_HIT_sum TCC_MISS_sum"
P4="FETCH_SIZE"
P5="WRITE_SIZE"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  ( cd $R && timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -o g -- "$@" > $OUT/p$i.log 2>&1 )
done
cd $R
python - "$OUT" "$FILTER" <<'PY'
import csv, glob, collections, sys
out, flt = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:80]
        if flt not in k: continue
        a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
import json
json.dump({k: dict({c: x[1] / max(x[0], 1) for c, x in d.items()}, dispatches=max(x[0] for x in d.values()))
           for k, d in agg.items()}, open(out + "/summary.json", "w"), indent=1)
for k, d in sorted(agg.items()):
    print(k)
    v = {c: x[1] / max(x[0], 1) for c, x in d.items()}
    print("   " + "  ".join(f"{c}={x:.4g}" for c, x in sorted(v.items())))
    if "SQ_BUSY_CYCLES" in v and "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("GRBM_GUI_ACTIVE"):
        # SQ_VALU_MFMA_BUSY_CYCLES: cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        # (= 8 x the kernel's shader-clock cycles: 5.46e6 for a 287 us kernel at 2.38 GHz)
        print(f"   MFMA busy = {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f} of 1024 SIMDs x kernel cycles")
    if "TCC_HIT_sum" in v:
        print(f"   L2 hit rate = {v['TCC_HIT_sum'] / max(v['TCC_HIT_sum'] + v['TCC_MISS_sum'], 1):.3f}")
    if "FETCH_SIZE" in v:
        print(f"   HBM read (FETCH_SIZE x2, gfx950 correction) = {2 * v['FETCH_SIZE'] / 1024:.1f} MiB/dispatch, WRITE_SIZE = {v.get('WRITE_SIZE', 0) / 1024:.1f} MiB/dispatch")
PY
