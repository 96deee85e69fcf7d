#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3l
mkdir -p $O
bash scripts/gpu_r3j.sh > $O/repair.txt 2>&1; cat $O/repair.txt
timeout 2400 python -m pytest tests/test_gpu_env_variants.py tests/test_gpu_patch_check.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
tail -5 $O/pytest.log
