#!/bin/bash
# Round 5: overlap-rule constants (real 1-rank RCCL) and the --gpus 4 bench line over the mock transport (functional, shared GPU)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05b
mkdir -p $O
PART=${PART:-rule}
if [ "$PART" = rule ]; then
  timeout 900 python scripts/overlap_rule.py --config cfg3 > $O/r05_overlap_rule.txt 2> $O/overlap_rule.err
  grep -v "amdgpu.ids" $O/overlap_rule.err | tail -5
  cat $O/r05_overlap_rule.txt
fi
if [ "$PART" = dist4 ]; then
  MOCK=$(python -c "from tests.mock_rccl.build import build; print(build())")
  ALFI_DIST_BACKEND=gloo ALFI_DIST_TRANSPORT=rccl ALFI_RCCL_LIB=$MOCK ALFI_BENCH_TIMEOUT_S=1500 timeout 1700 python bench.py --gpus 4 --steps 2 --warmup 1 > $O/r05_bench_dist4_native_transport_mock_sharedgpu_functional.json 2> $O/bench_dist4.err
  echo "exit $?"; tail -3 $O/bench_dist4.err
  python - $O/r05_bench_dist4_native_transport_mock_sharedgpu_functional.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step", "single_owner_levels_ms", "fcycle_ms", "fcycle_single_owner_levels_ms")})
print(d["per_rank"]["events_ms_fcycle"])
PY
fi
