#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in "ALFI_SPMV=legacy" "ALFI_NT=0" "ALFI_X=1"; do
  echo "=== $v"
  env $v python -m pytest tests/test_gpu_dist.py -m gpu -q -k "coarse-on-rank0 or P1FB" 2>&1 | grep -E "passed|failed|AssertionError: assert" | head -5
done
