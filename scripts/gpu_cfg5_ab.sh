#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_sv.py tests/test_gpu_parity.py -m gpu -x -q -k "p3 or large" 2>&1 | tail -3
for s in 1 0 2 4; do
  export ALFI_BIG_SPLIT=$s
  python bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('split $s', 'vps %.3f' % d['value'], 'apply GB/s %.0f' % d['roofline']['achieved'], 'finest %.0f' % d['roofline']['finest_level_GBps'], d['events_ms'])"
done
