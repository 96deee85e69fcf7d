#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3j
mkdir -p $O
ALFI_PATCH_CHECK_TOL=3e-9 python scripts/repair_check.py cfg4s 2>&1 | grep -v amdgpu.ids | tee $O/repair_cfg4s_tol3e-9.txt
ALFI_PATCH_CHECK_TOL=3e-9 ALFI_INVERT_MFMA=0 python scripts/repair_check.py cfg4s 2>&1 | grep -v amdgpu.ids | tee $O/repair_cfg4s_reg_tol3e-9.txt
ALFI_PATCH_CHECK_TOL=1e-9 python scripts/repair_check.py cfg3 2>&1 | grep -v amdgpu.ids | tee $O/repair_cfg3_tol1e-9.txt
