#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for w in 16 8 4; do
ALFI_COND_WAVES=$w python bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('waves=$w', d['value'], d['ms_per_step'], d['roofline']['finest_level_GBps'], d['roofline']['finest_level_avg_launch_us'])"
done
