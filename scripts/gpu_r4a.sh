#!/bin/bash
# round 4, first GPU contact: A/B of the interleaved small-patch layout and the vectorised BLAS-1 on configs 2 / 3, the
# headline config, then the GPU suite with durations
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4a
mkdir -p $O
B="python bench.py --no-cpu-baseline --steps 20 --warmup 3"
ALFI_BENCH_PROF=0 $B --config cfg2 > $O/cfg2_il.json 2> $O/cfg2_il.err
ALFI_BENCH_PROF=0 ALFI_PATCH_IL=0 $B --config cfg2 > $O/cfg2_noil.json 2> $O/cfg2_noil.err
ALFI_BENCH_PROF=0 $B --config cfg2 > $O/cfg2_il_b.json 2> $O/cfg2_il_b.err
ALFI_BENCH_PROF=0 $B --config cfg3 > $O/cfg3.json 2> $O/cfg3.err
$B --config cfg2 > $O/cfg2_events.json 2> $O/cfg2_events.err
$B --config cfg4 --steps 10 > $O/cfg4.json 2> $O/cfg4.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], d["ms_per_step"], d.get("rel_residual_after_timed_cycles"), (d.get("roofline") or {}).get("frac"),
              {k: round(v, 3) for k, v in (d.get("events_ms") or {}).items()} if isinstance(d.get("events_ms"), dict) else "")
    except Exception as e:
        print(f, "FAILED", e)
PY
timeout 2400 python -m pytest tests -x -q -m gpu --durations=40 > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
tail -60 $O/pytest.log
