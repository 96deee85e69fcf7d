#!/bin/bash
# CPU-side cost of the exchange points on a small level: 1-rank RCCL group, exchange points forced on, overlap on / off
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for cfg in cfg4t cfg4s; do
python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg plain       ms/cycle %.3f' % d['ms_per_step'])"
for ov in 1 0; do export ALFI_DIST_OVERLAP_MIN_DOFS=0
ALFI_DIST_FORCE=1 ALFI_DIST_MIN_DOFS=1000 ALFI_DIST_OVERLAP=$ov python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg forced ov=$ov ms/cycle %.3f' % d['ms_per_step'], d.get('events_ms_rank0'))"
done
done
