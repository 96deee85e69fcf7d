#!/bin/bash
# round 4: multiplicative sweep, pipelined product groups, three workgroups per CU (parity, stamps, variants); sigma chunk rows A/B on config 5
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4n
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_env_variants.py tests/test_frontend.py tests/test_gpu_dist.py -q -m gpu -x -k "persistent or multiplicative" > $O/pytest_mult.log 2>&1; tail -n 5 $O/pytest_mult.log
ALFI_HIP_LIB=$GRAFT_REPO_ROOT/alfi_amd/libalfi_hip_mtiming.so timeout 600 python scripts/mult_stamps.py cfg4 > $O/stamps.txt 2>&1
grep -v "amdgpu.ids" $O/stamps.txt | tail -n 12
for v in default mocc2 mg3 mu8 mocc4; do
  LIB=$GRAFT_REPO_ROOT/alfi_amd/libalfi_hip_$v.so
  [ $v = default ] && LIB=$GRAFT_REPO_ROOT/alfi_amd/libalfi_hip.so
  for pm in 1 0; do
    ALFI_MULT_PERSISTENT=$pm ALFI_HIP_LIB=$LIB timeout 600 python scripts/mult_time.py cfg4 > $O/mult_cfg4_${v}_p$pm.txt 2>&1
    echo "$v persistent=$pm: $(tail -n 2 $O/mult_cfg4_${v}_p$pm.txt | head -1)"
  done
done
for v in default sig128 sig64 default sig128 sig64; do
  LIB=$GRAFT_REPO_ROOT/alfi_amd/libalfi_hip_$v.so
  [ $v = default ] && LIB=$GRAFT_REPO_ROOT/alfi_amd/libalfi_hip.so
  ALFI_HIP_LIB=$LIB python bench.py --config cfg5 --no-cpu-baseline --steps 10 --warmup 3 > $O/cfg5_$v.json 2> $O/cfg5_$v.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/cfg5_$v.json") if l.startswith("{")][-1])
    print("cfg5 $v:", round(d["ms_per_step"], 3), "ms", d.get("rel_residual_after_timed_cycles"), "roofline", d["roofline"]["frac"], d["roofline"].get("achieved"))
except Exception as e:
    print("cfg5 $v FAILED", e)
PY
done
