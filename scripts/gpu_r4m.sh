#!/bin/bash
# round 4: inversion with the Newton reciprocal (parity + timing), bench lines of configs 2, 3, 4 with the workgroup-per-patch apply
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4m
mkdir -p $O
for v in invold default invold default; do
  LIB=$GRAFT_REPO_ROOT/alfi_amd/libalfi_hip_$v.so
  [ $v = default ] && LIB=$GRAFT_REPO_ROOT/alfi_amd/libalfi_hip.so
  ALFI_HIP_LIB=$LIB timeout 600 python scripts/factor_time.py cfg4s > $O/factor_cfg4s_$v.txt 2>&1
  echo "$v: $(grep -v amdgpu.ids $O/factor_cfg4s_$v.txt | tail -n 2 | tr '\n' ' ')"
done
timeout 1500 python -m pytest tests/test_gpu_patch_check.py tests/test_gpu_parity.py tests/test_gpu_newton.py tests/test_gpu_assemble.py tests/test_gpu_fullsize.py -q -m gpu -x > $O/pytest_inv.log 2>&1; tail -n 3 $O/pytest_inv.log
for C in cfg2 cfg3 cfg4 cfg2 cfg3 cfg4; do
  ALFI_BENCH_PROF=0 python bench.py --no-cpu-baseline --steps 20 --warmup 3 --config $C > $O/$C.json 2> $O/$C.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/$C.json") if l.startswith("{")][-1])
    print("$C:", round(d["ms_per_step"], 3), "ms", d.get("rel_residual_after_timed_cycles"))
except Exception as e:
    print("$C FAILED", e)
PY
done
python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/cfg4_prof.json 2> $O/cfg4_prof.err; head -c 700 $O/cfg4_prof.json; echo
python scripts/factor_time.py cfg4 > $O/factor_cfg4.txt 2>&1; grep -v amdgpu.ids $O/factor_cfg4.txt | tail -n 2
