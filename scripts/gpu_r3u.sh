#!/bin/bash
# condensed apply: tests, then kernel traces of the three-launch forms (per patch: ALFI_COND_SPLIT=2; per chunk of groups: 1)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_condensed.py tests/test_gpu_sv.py tests/test_gpu_env_variants.py -x -q -m gpu -k "not smoother_paths" 2>&1 | tail -5
bash scripts/gpu_r3t.sh ALFI_COND_SPLIT=2 - ALFI_COND_SPLIT=2 - ALFI_COND_SPLIT=0
