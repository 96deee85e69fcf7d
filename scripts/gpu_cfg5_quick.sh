#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_condensed.py -m gpu -x -q 2>&1 | tail -3
python bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_cfg5_c1.json 2> gpurun_out/bench_cfg5_c1.err
python -c "
import json
d = json.load(open('gpurun_out/bench_cfg5_c1.json'))
print({k: d.get(k) for k in ['value', 'ms_per_step', 'rel_residual_after_timed_cycles', 'patch_factor_GB']})
print(d['roofline']['finest_level_GBps'], d['roofline']['finest_level_avg_launch_us']); print(d['events_ms'])"
