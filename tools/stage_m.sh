#!/bin/bash
# round 4, stage M: fused-epilogue projections as two 4-wave workgroups per CU (store phases alternate) vs one 8-wave
R=$PWD; O=$R/gpurun_out/stage_m; mkdir -p $O
{ echo "8 waves"; python tools/epilogue_probe.py 2>&1 | tail -4
  echo "4 waves, LDS-DMA"; PDN_ROWRES_EPI_NW=4 PDN_ROWRES_EPI_STAGE=0 python tools/epilogue_probe.py 2>&1 | tail -4
  echo "4 waves, register staged"; PDN_ROWRES_EPI_NW=4 PDN_ROWRES_EPI_STAGE=1 python tools/epilogue_probe.py 2>&1 | tail -4; } | tee $O/probe.txt
