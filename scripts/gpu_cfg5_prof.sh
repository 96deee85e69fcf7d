#!/bin/bash
# kernel trace of the config-5 bench (setup kernels of the large-patch path + the cycle)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_cfg5
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg5 -- python3 bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_cfg5_rocprof.json 2> gpurun_out/prof_cfg5.err
echo "rocprof exit $?"
python - <<PY
import json, glob, csv
d = json.load(open("gpurun_out/bench_cfg5_rocprof.json"))
print("vps %.3f" % d["value"], d["setup_s"])
f = glob.glob("gpurun_out/prof_cfg5/**/*_kernel_stats.csv", recursive=True)[0]
for i, row in enumerate(csv.reader(open(f))):
    if i < 9: print(row[0][:60], row[1:5])
PY
