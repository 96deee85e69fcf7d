#!/bin/bash
# config 5 bench line again (reads the refreshed HBM counters of the three-launch apply), then config 5L
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03f
mkdir -p $O
python bench.py --config cfg5 --steps 10 --warmup 3 > $O/r03_bench_cfg5.json 2> $O/bench_cfg5.err
head -c 300 $O/r03_bench_cfg5.json; echo
PART=cfg5L bash scripts/gpu_r3_final.sh
ALFI_COND_SPLIT=0 python bench.py --config cfg5L --steps 5 --warmup 2 --no-cpu-baseline > $O/ab_cfg5L_split0.json 2> $O/ab_cfg5L_split0.err
python - $O/ab_cfg5L_split0.json $O/r03_bench_cfg5L.json <<'PY' | tee $O/r03_cond_apply_ab_cfg5L.txt
import json, sys
for f, v in zip(sys.argv[1:], ("0", "1")):
    d = json.load(open(f)); r = d["roofline"]
    print("ALFI_COND_SPLIT=%s  %s: %.2f ms/V-cycle without events, finest-level apply %.1f us = %.0f GB/s, all levels %.0f GB/s"
          % (v, r["kernel"], d["ms_per_step_without_events"], r["finest_level_avg_launch_us"], r["finest_level_GBps"], r["achieved"]))
PY
