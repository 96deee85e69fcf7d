#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3k
mkdir -p $O
bash scripts/gpu_r3j.sh > $O/repair.txt 2>&1; cat $O/repair.txt
timeout 2400 python -m pytest tests/test_gpu_env_variants.py tests/test_gpu_parity.py tests/test_gpu_sv.py tests/test_gpu_condensed.py tests/test_frontend.py tests/test_gpu_outer.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
tail -5 $O/pytest.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
for C in cfg3 cfg4 cfg5 cfg2; do
  ALFI_BENCH_PROF=0 $B --config $C > $O/${C}_new.json 2> $O/${C}_new.err
  ALFI_BENCH_PROF=0 ALFI_FUSED_ALL=0 $B --config $C > $O/${C}_fusedall0.json 2> $O/${C}_fusedall0.err
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), "FAILED", open(f[:-5] + ".err").read()[-600:]); continue
    print("%-22s ms/step %8.3f noev %8.3f res %.2e" % (os.path.basename(f), d["ms_per_step"], d["ms_per_step_without_events"], d["rel_residual_after_timed_cycles"]))
PY
