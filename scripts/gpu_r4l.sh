#!/bin/bash
# round 4: multiplicative sweep with products in LDS + row sums (parity, stamps, variants); inversion kernel with batched panel reads
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4l
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_env_variants.py tests/test_frontend.py tests/test_gpu_dist.py -q -m gpu -x -k "persistent or multiplicative" > $O/pytest_mult.log 2>&1; tail -n 5 $O/pytest_mult.log
ALFI_HIP_LIB=$GRAFT_REPO_ROOT/alfi_amd/libalfi_hip_mtiming.so timeout 600 python scripts/mult_stamps.py cfg4 > $O/stamps.txt 2>&1
grep -v "amdgpu.ids" $O/stamps.txt | tail -n 12
for v in default mu40 mu24 mg4 mg3; do
  LIB=$GRAFT_REPO_ROOT/alfi_amd/libalfi_hip_$v.so
  [ $v = default ] && LIB=$GRAFT_REPO_ROOT/alfi_amd/libalfi_hip.so
  for pm in 1 0; do
    ALFI_MULT_PERSISTENT=$pm ALFI_HIP_LIB=$LIB timeout 600 python scripts/mult_time.py cfg4 > $O/mult_cfg4_${v}_p$pm.txt 2>&1
    echo "$v persistent=$pm: $(tail -n 2 $O/mult_cfg4_${v}_p$pm.txt | head -1)"
  done
done
# inversion
for v in invold default invold default; do
  LIB=$GRAFT_REPO_ROOT/alfi_amd/libalfi_hip_$v.so
  [ $v = default ] && LIB=$GRAFT_REPO_ROOT/alfi_amd/libalfi_hip.so
  OMP_NUM_THREADS=1 ALFI_HOST_THREADS=1 ALFI_HIP_LIB=$LIB timeout 600 python scripts/factor_time.py cfg4s $O/inv_$v.npy > $O/factor_cfg4s_$v.txt 2>&1
  echo "$v: $(grep -v amdgpu.ids $O/factor_cfg4s_$v.txt | tail -n 2 | tr '\n' ' ')"
done
python - <<PY
import numpy as np
a = np.load("$O/inv_invold.npy"); b = np.load("$O/inv_default.npy")
print("apply with the new inverses", "bitwise equal to the old kernel's" if np.array_equal(a, b) else "DIFFERS %.3e" % (np.abs(a - b).max() / np.abs(a).max()))
PY
rm -f $O/*.npy
ALFI_HIP_LIB=$GRAFT_REPO_ROOT/alfi_amd/libalfi_hip_inv2timing.so timeout 600 python scripts/invert_phases.py cfg4s > $O/invert_phases.txt 2>&1
grep -v "amdgpu.ids" $O/invert_phases.txt | tail -n 18
timeout 900 python -m pytest tests/test_gpu_patch_check.py tests/test_gpu_parity.py -q -m gpu -x > $O/pytest_inv.log 2>&1; tail -n 3 $O/pytest_inv.log
