#!/bin/bash
# round 3: parity of the one-workgroup smoother and the de-duplicated SpMV, then same-box A/B timings
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3b
mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_env_variants.py tests/test_gpu_parity.py tests/test_golden.py tests/test_frontend.py -x -q > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
tail -15 $O/pytest.log
run() { # name, env..., -- args
  name=$1; shift
  env "$@" > /dev/null 2>&1
}
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
ALFI_BENCH_PROF=0 $B --config cfg2 > $O/cfg2_tiny.json 2> $O/cfg2_tiny.err
ALFI_BENCH_PROF=0 ALFI_TINY_BYTES=0 $B --config cfg2 > $O/cfg2_notiny.json 2> $O/cfg2_notiny.err
ALFI_BENCH_PROF=0 ALFI_TINY_BYTES=1500000 $B --config cfg2 > $O/cfg2_tiny1p5.json 2> $O/cfg2_tiny1p5.err
ALFI_BENCH_PROF=0 ALFI_TINY_BYTES=12000000 $B --config cfg2 > $O/cfg2_tiny12.json 2> $O/cfg2_tiny12.err
ALFI_BENCH_PROF=0 $B --config cfg3 > $O/cfg3_new.json 2> $O/cfg3_new.err
ALFI_BENCH_PROF=0 ALFI_SPMV_DEDUP=0 ALFI_FUSED_REDUCE_MAX=256 ALFI_TINY_BYTES=0 $B --config cfg3 > $O/cfg3_old.json 2> $O/cfg3_old.err
ALFI_BENCH_PROF=0 ALFI_SPMV_DEDUP=0 $B --config cfg3 > $O/cfg3_nodedup.json 2> $O/cfg3_nodedup.err
$B --config cfg4 > $O/cfg4_dedup.json 2> $O/cfg4_dedup.err
ALFI_SPMV_DEDUP=0 $B --config cfg4 > $O/cfg4_nodedup.json 2> $O/cfg4_nodedup.err
ALFI_XCD_MAP=1 $B --config cfg4 > $O/cfg4_dedup_xcd.json 2> $O/cfg4_dedup_xcd.err
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), "FAILED", open(f[:-5] + ".err").read()[-600:]); continue
    print("%-22s ms/step %8.3f noev %8.3f spmv %7.1f GB/s (%.1f us) apply frac %.3f res %.2e" % (os.path.basename(f), d["ms_per_step"], d["ms_per_step_without_events"],
          d["spmv_finest"]["achieved_GBps"], d["spmv_finest"]["avg_launch_us"], d["roofline"]["frac"], d["rel_residual_after_timed_cycles"]))
PY
