#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
for v in 1 0; do
ALFI_FUSED_SMOOTHER=$v ALFI_BENCH_PROF=0 python bench.py --config cfg2 --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('cfg2 fused_smoother=$v ms/cycle %.3f res %.3e' % (d['ms_per_step'], d['rel_residual_after_timed_cycles']))"
done
for v in 1 0; do
ALFI_FUSED_REDUCE=$v ALFI_BENCH_PROF=0 python bench.py --config cfg4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('cfg4 fused_reduce=$v ms/cycle %.3f res %.3e' % (d['ms_per_step'], d['rel_residual_after_timed_cycles']), d['events_ms']['BLAS1'])"
done
