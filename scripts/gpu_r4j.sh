#!/bin/bash
# round 4: additive apply tail A/B at kernel level; phases of the inversion kernel
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4j
mkdir -p $O
ALFI_HIP_LIB=$GRAFT_REPO_ROOT/alfi_amd/libalfi_hip_invtiming.so timeout 600 python scripts/invert_phases.py cfg4s > $O/invert_phases.txt 2>&1
grep -v "amdgpu.ids" $O/invert_phases.txt | tail -n 20
for C in cfg3 cfg4; do
  for T in 0 8192 -1 0 8192 -1 16384 2048; do
    echo "== $C tail $T"
    ALFI_APPLY_TAIL=$T timeout 600 python scripts/apply_time.py $C 2>&1 | grep "level" | head -2
  done
done > $O/apply_tail.txt 2>&1
cat $O/apply_tail.txt
