#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3i
mkdir -p $O
START=$(date +%s)
timeout 2400 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_sv.py tests/test_gpu_condensed.py tests/test_gpu_mms.py tests/test_gpu_coarse_sparse.py tests/test_gpu_outer.py tests/test_gpu_graddiv.py tests/test_golden.py tests/test_ref_kernels.py -x -q -m gpu --durations=8 > $O/pytest_rest.log 2>&1
echo "pytest exit $? in $(( $(date +%s) - START )) s" >> $O/pytest_rest.log
tail -16 $O/pytest_rest.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
ALFI_BENCH_PROF=0 $B --config cfg3 > $O/cfg3_base.json 2> $O/cfg3_base.err
ALFI_BENCH_PROF=0 ALFI_SMALL_LEVEL_WG=1000 $B --config cfg3 > $O/cfg3_wg1000.json 2> $O/cfg3_wg1000.err
ALFI_BENCH_PROF=0 ALFI_SMALL_LEVEL_WG=200 $B --config cfg3 > $O/cfg3_wg200.json 2> $O/cfg3_wg200.err
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), "FAILED", open(f[:-5] + ".err").read()[-600:]); continue
    print("%-22s ms/step %8.3f noev %8.3f res %.2e" % (os.path.basename(f), d["ms_per_step"], d["ms_per_step_without_events"], d["rel_residual_after_timed_cycles"]))
PY
