#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3f
mkdir -p $O
for C in cfg4s cfg3; do
  python scripts/factor_time.py $C > $O/factor_${C}_mfma.txt 2>&1
  ALFI_INVERT_MFMA=0 python scripts/factor_time.py $C > $O/factor_${C}_reg.txt 2>&1
done
for f in $O/factor_*.txt; do echo $f; tail -n 1 $f; done
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_patch_check.py tests/test_gpu_dist.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
tail -8 $O/pytest.log
