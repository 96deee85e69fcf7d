#!/bin/bash
# round 4: additive apply with the level's last patches a workgroup each: parity, then A/B of the tail length on configs 3 and 4
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4h
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_dist.py -q -m gpu -x > $O/pytest.log 2>&1; tail -n 5 $O/pytest.log
for C in cfg3 cfg4; do
  for T in 0 8192 -1 0 8192 -1 16384 4096; do
    ALFI_BENCH_PROF=0 ALFI_APPLY_TAIL=$T python bench.py --no-cpu-baseline --steps 20 --warmup 3 --config $C > $O/${C}_tail$T.json 2> $O/${C}_tail$T.err
    python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/${C}_tail$T.json") if l.startswith("{")][-1])
    print("$C tail $T:", round(d["ms_per_step"], 3), "ms", d.get("rel_residual_after_timed_cycles"), d["roofline"]["frac"])
except Exception as e:
    print("$C tail $T FAILED", e)
PY
  done
done
