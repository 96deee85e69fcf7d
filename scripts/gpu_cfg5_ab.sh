#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_sv.py tests/test_gpu_parity.py tests/test_frontend.py -m gpu -x -q -k "p3 or large or macro" 2>&1 | tail -3
python bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline --verbose 2>gpurun_out/cfg5_verbose.err | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('no profiler: vps %.3f' % d['value'], d['setup_s'])"
grep -i "setup\|factor\|level\|s)" gpurun_out/cfg5_verbose.err | tail -20
