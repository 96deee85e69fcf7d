#!/bin/bash
# round 4: multiplicative sweep with a workgroup per patch: parity tests first, then waves per patch / loads in flight A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4g
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_env_variants.py tests/test_frontend.py tests/test_gpu_dist.py -q -m gpu -x -k "persistent or multiplicative" > $O/pytest_mult.log 2>&1; tail -n 5 $O/pytest_mult.log
for v in default w8 w4u32 w8u8 w8r2 w16r2 w2; do
  LIB=$GRAFT_REPO_ROOT/alfi_amd/libalfi_hip_$v.so
  [ $v = default ] && LIB=$GRAFT_REPO_ROOT/alfi_amd/libalfi_hip.so
  for pm in 1 0; do
    ALFI_MULT_PERSISTENT=$pm ALFI_HIP_LIB=$LIB timeout 600 python scripts/mult_time.py cfg4 > $O/mult_cfg4_${v}_p$pm.txt 2>&1
    echo "$v persistent=$pm: $(tail -n 2 $O/mult_cfg4_${v}_p$pm.txt | head -1)"
  done
done
grep -h checksum $O/mult_cfg4_*_p*.txt
