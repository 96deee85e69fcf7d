#!/bin/bash
# condensed apply: tests, same-box A/B (one launch / three launches), kernel traces of the three-launch form
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
bash scripts/gpu_r3s.sh
bash scripts/gpu_r3t.sh - ALFI_COND_RU=16 ALFI_COND_WAVES=16 ALFI_NT=0
