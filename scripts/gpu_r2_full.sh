#!/bin/bash
# full GPU regression + the three BASELINE benches
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -12 gpurun_out/pytest_gpu.log
for cfg in cfg2 cfg3; do
ALFI_BENCH_PROF=0 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null > gpurun_out/bench_$cfg.json
python -c "
import sys, json
d = json.load(open('gpurun_out/bench_$cfg.json')); print('$cfg ms/cycle %.3f' % d['ms_per_step'], d['events_ms'])"
done
python bench.py > gpurun_out/bench_cfg4.json 2> gpurun_out/bench_cfg4.err
tail -2 gpurun_out/bench_cfg4.err
python -c "
import json
d = json.load(open('gpurun_out/bench_cfg4.json'))
for k in ['value', 'ms_per_step', 'roofline', 'spmv_finest', 'cpu_baseline', 'setup_s', 'events_ms']: print(k, d.get(k))"
