#!/bin/bash
# round 4: where the time of the persistent sweep goes (clock stamps), occupancy variants
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4i
mkdir -p $O
for v in timing w8timing; do
  ALFI_HIP_LIB=$GRAFT_REPO_ROOT/alfi_amd/libalfi_hip_$v.so timeout 600 python scripts/mult_stamps.py cfg4 > $O/stamps_$v.txt 2>&1
  grep -v "amdgpu.ids" $O/stamps_$v.txt | tail -n 12
done
for v in default r2u8 r2u8o4 w8r2u8o4; do
  LIB=$GRAFT_REPO_ROOT/alfi_amd/libalfi_hip_$v.so
  [ $v = default ] && LIB=$GRAFT_REPO_ROOT/alfi_amd/libalfi_hip.so
  for pm in 1 0; do
    ALFI_MULT_PERSISTENT=$pm ALFI_HIP_LIB=$LIB timeout 600 python scripts/mult_time.py cfg4 > $O/mult_cfg4_${v}_p$pm.txt 2>&1
    echo "$v persistent=$pm: $(tail -n 2 $O/mult_cfg4_${v}_p$pm.txt | head -1)"
  done
done
