#!/bin/bash
# SpMV: strip-wise XCD mapping of the de-duplicated kernel (A/B, same box) + HBM traffic of the best variant
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3m
mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
for S in 0 16 64 256 1024; do
  ALFI_XCD_MAP=$S $B > $O/cfg4_xcd$S.json 2> $O/cfg4_xcd$S.err
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), "FAILED", open(f[:-5] + ".err").read()[-600:]); continue
    print("%-22s ms/step %8.3f spmv %7.1f GB/s (%.1f us) apply %.3f res %.2e" % (os.path.basename(f), d["ms_per_step"],
          d["spmv_finest"]["achieved_GBps"], d["spmv_finest"]["avg_launch_us"], d["roofline"]["frac"], d["rel_residual_after_timed_cycles"]))
PY
