#!/bin/bash
# condensed apply (config 5): one launch per apply (ALFI_COND_SPLIT=0) against front / sigma / back -- same-box A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3s
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_condensed.py tests/test_gpu_sv.py -x -q -m gpu 2>&1 | tail -5
for v in "0 8" "1 8" "0 8" "1 8"; do
  set -- $v
  ALFI_COND_SPLIT=$1 ALFI_COND_WAVES=$2 python bench.py --config cfg5 --steps 10 --warmup 3 --no-cpu-baseline > $O/b_$1_$2.json 2> $O/b_$1_$2.err
  python - $O/b_$1_$2.json "$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(sys.argv[2], "ms/cycle", round(d["ms_per_step_without_events"], 2), "finest GB/s", round(r["finest_level_GBps"]), "finest us", round(r["finest_level_avg_launch_us"], 1), "all", round(r["achieved"]))
PY
done
