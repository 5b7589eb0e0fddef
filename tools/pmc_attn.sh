#!/bin/bash
# PMC passes over the attention kernels; prints per-kernel averages.
R=$PWD; OUT=$R/gpurun_out/pmc_attn; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"
P3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -o g -- python $R/tools/attn_one.py 3 128 > $OUT/p$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("gpurun_out/pmc_attn/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]
        if "attention" not in k: continue
        a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in agg.items():
    print(k)
    print("   " + "  ".join(f"{c}={v[1]/max(v[0],1):.3g}" for c, v in sorted(d.items())))
PY
