#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3d
mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_env_variants.py tests/test_gpu_parity.py tests/test_gpu_patch_check.py tests/test_frontend.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
tail -8 $O/pytest.log
for C in cfg4s cfg3; do
  python scripts/factor_time.py $C > $O/factor_${C}_mfma.txt 2>&1
  ALFI_INVERT_MFMA=0 python scripts/factor_time.py $C > $O/factor_${C}_reg.txt 2>&1
done
tail -2 $O/factor_*.txt
