#!/bin/bash
# round 4: additive apply, a workgroup per patch: parity, waves per patch A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4k
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sv.py tests/test_gpu_dist.py -q -m gpu -x > $O/pytest.log 2>&1; tail -n 5 $O/pytest.log
for C in cfg3 cfg4; do
  for v in default aw2 aw8 default aw2 aw8; do
    LIB=$GRAFT_REPO_ROOT/alfi_amd/libalfi_hip_$v.so
    [ $v = default ] && LIB=$GRAFT_REPO_ROOT/alfi_amd/libalfi_hip.so
    echo "== $C $v"
    ALFI_HIP_LIB=$LIB timeout 600 python scripts/apply_time.py $C 2>&1 | grep "level" | head -2
  done
done > $O/apply_w.txt 2>&1
cat $O/apply_w.txt
