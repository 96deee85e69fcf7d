#!/bin/bash
# HBM traffic of the three launches of the condensed apply, kernel by kernel (config 5)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3p
mkdir -p $O
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -- python3 $GRAFT_REPO_ROOT/bench.py --config cfg5 --no-cpu-baseline --steps 1 --warmup 0 > $O/f.json 2> $O/f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -- python3 $GRAFT_REPO_ROOT/bench.py --config cfg5 --no-cpu-baseline --steps 1 --warmup 0 > $O/w.json 2> $O/w.err
cd $GRAFT_REPO_ROOT
for K in cond_gfront_kernel cond_gsigma_kernel cond_gback_kernel; do
  python scripts/pmc_summary.py $O/f $O/w "void $K" $O/pmc_$K.json "$K finest level" 0 max | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$K', 'fetch GB', round(d['fetch_bytes_per_launch']/1e9,4), 'write GB', round(d['write_bytes_per_launch']/1e9,4), 'launches', d['launches'])"
done
rm -rf $O/f $O/w
