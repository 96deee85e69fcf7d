#!/bin/bash
# after the condensed-apply work: the GPU tests that touch condensed factors (single GPU and partitioned), then config 5
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_condensed.py tests/test_gpu_sv.py tests/test_gpu_fullsize.py tests/test_gpu_patch_check.py tests/test_gpu_env_variants.py -x -q -m gpu -k "not smoother_paths" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "sv or SV or condensed or macro" 2>&1 | tail -3
PART=cfg5 bash scripts/gpu_r3_final.sh
