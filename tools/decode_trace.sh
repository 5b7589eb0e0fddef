#!/bin/bash
# Builds a -DDEC_TRACE copy of the library (phase timestamps inside the decode kernels, decode_stage.h) next to the
# shipped one and prints where a decode token's time goes:  bash tools/decode_trace.sh   (on the GPU box)
# The traced library is built HERE (hipcc cross-compiles) into tools/micro/libpdnhip_trace.bin when it is missing.
set -e
R=$(cd $(dirname $0)/.. && pwd)
T=$R/tools/micro/libpdnhip_trace.bin
if [ ! -f $T ] || [ "$1" = build ]; then
  B=$(mktemp -d)
  for f in $R/pydynet_amd/csrc/*.hip; do
    n=$(basename $f .hip)
    case $n in decode|decode_layer|decode_block) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDEC_TRACE -mllvm -amdgpu-kernarg-preload-count=16 -I$R/pydynet_amd/csrc -c $f -o $B/$n.o ;;
      *) cp $R/pydynet_amd/csrc/build/$n.o $B/$n.o ;; esac
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $B/*.o -o $T
  rm -rf $B
  [ "$1" = build ] && exit 0
fi
PDN_LIB=$T python $R/tools/decode_trace.py
