#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3e
mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_assemble.py tests/test_gpu_parity.py tests/test_gpu_patch_check.py tests/test_gpu_newton.py tests/test_gpu_env_variants.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
tail -8 $O/pytest.log
for C in cfg4s cfg3; do
  python scripts/factor_time.py $C > $O/factor_${C}_mfma.txt 2>&1
  ALFI_INVERT_MFMA=0 python scripts/factor_time.py $C > $O/factor_${C}_reg.txt 2>&1
done
for f in $O/factor_*.txt; do echo $f; tail -n 1 $f; done
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
ALFI_BENCH_PROF=0 $B --config cfg2 > $O/cfg2_tiny1M.json 2> $O/cfg2_tiny1M.err
ALFI_BENCH_PROF=0 ALFI_TINY_BYTES=0 $B --config cfg2 > $O/cfg2_notiny.json 2> $O/cfg2_notiny.err
ALFI_BENCH_PROF=0 ALFI_TINY_BYTES=3500000 $B --config cfg2 > $O/cfg2_tiny3p5M.json 2> $O/cfg2_tiny3p5M.err
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), "FAILED", open(f[:-5] + ".err").read()[-600:]); continue
    print("%-22s ms/step %8.3f noev %8.3f other %8.3f res %.2e ev %s" % (os.path.basename(f), d["ms_per_step"], d["ms_per_step_without_events"],
          d["other_restriction_setting"]["ms_per_step"], d["rel_residual_after_timed_cycles"], d["events_ms"].get("KSP_TINY")))
PY
