#!/bin/bash
# kernel trace of one config without HIP events: $CFG (default cfg2); summary -> gpurun_out/prof_$CFG/
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
CFG=${CFG:-cfg2}
mkdir -p gpurun_out/prof_$CFG
cd /tmp
ALFI_BENCH_PROF=0 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$CFG -o run -- python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$CFG/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_$CFG/bench.err
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_$CFG -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -40 {}'
# keep the per-dispatch trace small: aggregate by (kernel, grid size)
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/prof_$CFG/**/*kernel_trace.csv", recursive=True)
if f:
    agg = collections.defaultdict(lambda: [0, 0.0])
    with open(f[0]) as fh:
        for r in csv.DictReader(fh):
            k = (r["Kernel_Name"][:60], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?"))
            agg[k][0] += 1
            agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    with open("gpurun_out/prof_$CFG/by_grid.txt", "w") as out:
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            out.write("%-62s grid %-10s n %6d total %10.1f us avg %8.2f us\n" % (k[0], k[1], v[0], v[1], v[1] / v[0]))
    import os
    os.remove(f[0])
PY
head -60 gpurun_out/prof_$CFG/by_grid.txt
