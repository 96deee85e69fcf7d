#!/bin/bash
# HBM traffic of the three launches of a condensed apply on config 5's finest level, kernel by kernel: separate --pmc passes of
# bench.py --config cfg5 --steps 1 --warmup 0, summarised per kernel (largest grid = the finest level) with provenance.
# usage (GPU box): ALFI_COMMIT=<hash> bash scripts/gpu_cfg5_pmc.sh   -> gpurun_out/cfg5_pmc/r04_pmc_cond_{front,sigma,back}_cfg5.json
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/cfg5_pmc
mkdir -p $O
STAMP=$(date -u +%Y-%m-%dT%H:%MZ)
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc5_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --config cfg5 --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc5_fetch.json 2> $O/pmc5_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc5_write -- python3 $GRAFT_REPO_ROOT/bench.py --config cfg5 --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc5_write.json 2> $O/pmc5_write.err
cd $GRAFT_REPO_ROOT
for k in front sigma back; do
  python scripts/pmc_summary.py $O/pmc5_fetch $O/pmc5_write "void cond_${k}_kernel" $O/r04_pmc_cond_${k}_cfg5.json "r04 ($STAMP) cond_${k}_kernel, config 5's finest level (largest grid), all launches of the run" 0 max
done
python scripts/level_report.py cfg5 > $O/level_report_cfg5.txt 2>&1 || true
rm -rf $O/pmc5_fetch $O/pmc5_write
ls -la $O
