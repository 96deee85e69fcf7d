#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3o
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log; tail -3 $O/pytest.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
ALFI_BENCH_PROF=0 $B --config cfg2 > $O/cfg2_a.json 2> $O/cfg2_a.err
ALFI_BENCH_PROF=0 $B --config cfg2 > $O/cfg2_b.json 2> $O/cfg2_b.err
ALFI_BENCH_PROF=0 $B --config cfg1 > $O/cfg1.json 2> $O/cfg1.err
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace -d $O/trace_cfg2 -o run -- python3 $GRAFT_REPO_ROOT/scripts/trace_cycle.py --config cfg2 --cycles 6 > $O/trace_cfg2.out 2> $O/trace_cfg2.err
cd $GRAFT_REPO_ROOT
DB=$(find $O/trace_cfg2 -name "*.db" | head -1)
python scripts/timeline_summary.py $DB 4 > $O/timeline_cfg2.txt 2>&1
rm -rf $O/trace_cfg2
head -14 $O/timeline_cfg2.txt | cut -c1-150
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*.json")):
    d = json.load(open(f))
    print("%-22s ms/step %8.3f noev %8.3f res %.2e" % (os.path.basename(f), d["ms_per_step"], d["ms_per_step_without_events"], d["rel_residual_after_timed_cycles"]))
PY
