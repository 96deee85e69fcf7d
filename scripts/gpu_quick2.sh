#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in cfg4t cfg2; do
for pr in 1 2 0; do
ALFI_BENCH_PROF=$pr python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg prof=$pr ms/cycle %.3f' % d['ms_per_step'])"
done
done
