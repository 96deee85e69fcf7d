#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 3300 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_all.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu_all.log
grep -E "^FAILED|^ERROR|passed|failed|pytest exit" gpurun_out/pytest_gpu_all.log | tail -30
