#!/bin/bash
# rest of the GPU suite after the env-variant fix + same-box A/B of the fused smoother
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -12 gpurun_out/pytest_gpu.log
for c in 3d-P2FB 2d-all-distributed; do echo $c; python scripts/ab_cycle.py $c 2>&1 | tail -5; done
for cfg in cfg2 cfg3; do
for v in 1 0; do
ALFI_FUSED_SMOOTHER=$v ALFI_BENCH_PROF=0 python bench.py --config $cfg --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null > gpurun_out/bench_${cfg}_f$v.json
python -c "
import sys, json
d = json.load(open('gpurun_out/bench_${cfg}_f$v.json')); print('$cfg fused=$v ms/cycle %.3f res %.3e' % (d['ms_per_step'], d['rel_residual_after_timed_cycles']), d['events_ms'])"
done
done
