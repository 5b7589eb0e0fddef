#!/bin/bash
# Builds a -DATT_TRACE copy of the library (per-wave phase timestamps inside the attention forward kernel) and prints
# where a workgroup's life goes:  bash tools/attn_trace.sh [build]    (run on the GPU box; `build` only builds, here)
set -e
R=$(cd $(dirname $0)/.. && pwd)
T=$R/tools/micro/libpdnhip_atttrace.bin
if [ ! -f $T ] || [ "$1" = build ]; then
  B=$(mktemp -d)
  for f in $R/pydynet_amd/csrc/*.hip; do
    n=$(basename $f .hip)
    case $n in attention) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DATT_TRACE -I$R/pydynet_amd/csrc -c $f -o $B/$n.o ;;
      *) cp $R/pydynet_amd/csrc/build/$n.o $B/$n.o ;; esac
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $B/*.o -o $T
  rm -rf $B
  [ "$1" = build ] && exit 0
fi
PDN_LIB=$T python $R/tools/attn_trace.py
