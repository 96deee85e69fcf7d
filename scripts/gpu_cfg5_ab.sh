#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in 1 0; do
ALFI_CONDENSE=$v python bench.py --config ${CFG:-cfg5} --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${CFG:-cfg5}_c$v.json 2> gpurun_out/bench_${CFG:-cfg5}_c$v.err
tail -2 gpurun_out/bench_${CFG:-cfg5}_c$v.err
python -c "
import json
d = json.load(open('gpurun_out/bench_${CFG:-cfg5}_c$v.json'))
print('condense=$v', {k: d.get(k) for k in ['value', 'ms_per_step', 'rel_residual_after_timed_cycles', 'patch_factor_GB', 'patch_factor_bytes_per_dof_finest', 'setup_s', 'vcycle_hbm_frac_of_peak']})
print(d['roofline']); print(d['events_ms'])"
done
