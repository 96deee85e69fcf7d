#!/bin/bash
# A/B runs of bench.py under environment variants on the same box: VARIANTS="name:ENV1=a,ENV2=b name2:..."
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in $VARIANTS; do
  name=${v%%:*}; envs=$(echo "${v#*:}" | tr ',' ' ')
  env $envs python bench.py --no-cpu-baseline ${BENCH_ARGS:---steps 5 --warmup 2} > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$name.json"))
print("$name", "ms/step %.1f" % d["ms_per_step"], "apply %.0f GB/s" % d["roofline"]["finest_level_GBps"], "spmv %.0f GB/s" % d["spmv_finest"]["achieved_GBps"], d["events_ms"], d["config"].get("wavefronts_per_sweep"), d["rel_residual_after_timed_cycles"])
PY
done
