#!/bin/bash
# A second libalfi_hip.so with extra compile flags, for same-box A/B runs:  scripts/build_variant.sh NAME -DALFI_X=0 ...
# -> alfi_amd/variants/libalfi_hip_NAME.so (git-ignored, travels with gpurun); select it with ALFI_HIP_LIB=<path>.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
OBJ=/tmp/alfi_variant_$NAME
mkdir -p $OBJ $ROOT/alfi_amd/variants
pids=()
for f in $ROOT/alfi_amd/csrc/*.hip; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -I $ROOT/include -I $ROOT/alfi_amd/csrc "$@" -c $f -o $OBJ/$(basename $f .hip).o &
  pids+=($!)
  if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o $ROOT/alfi_amd/variants/libalfi_hip_$NAME.so $OBJ/*.o
ls -la $ROOT/alfi_amd/variants/libalfi_hip_$NAME.so
