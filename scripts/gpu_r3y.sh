#!/bin/bash
# configs 2 and 3: same-box A/B of the SpMV switches (the de-duplicated kernel and its XCD map were tuned on config 4)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3y
mkdir -p $O
for C in cfg3 cfg2; do
for v in "-" "ALFI_SPMV_DEDUP=0" "ALFI_XCD_MAP=1" "ALFI_XCD_MAP=0" "ALFI_XCD_MAP=16" "ALFI_XCD_MAP=256" "-"; do
  [ "$v" = "-" ] && v=""
  for kv in $v; do export "$kv"; done
  python bench.py --config $C --steps 30 --warmup 10 --no-cpu-baseline > $O/b.json 2> $O/b.err
  for kv in $v; do unset "${kv%%=*}"; done
  python - $O/b.json "$C $v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("%-28s %.3f ms/cycle without events (%.3f with), spmv finest %.1f us = %.0f GB/s, apply finest %.1f us"
      % (sys.argv[2], d["ms_per_step_without_events"], d["ms_per_step"], d["spmv_finest"]["avg_launch_us"], d["spmv_finest"]["achieved_GBps"], d["roofline"]["finest_level_avg_launch_us"]))
PY
done
done
