#!/bin/bash
# Round 5, first GPU check of the cell-centric operator refresh: parity tests, then Newton-step timings and kernel stats at config-4 size.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05a
mkdir -p $O
PART=${PART:-tests}
if [ "$PART" = tests ]; then
  timeout 900 python -m pytest tests/test_gpu_assemble.py -x -q 2>&1 | tail -15 > $O/tests.txt
  cat $O/tests.txt
fi
if [ "$PART" = newton ]; then
  timeout 600 python scripts/newton_step_time.py cfg4 --re 10 100 1000 > $O/newton_cfg4.txt 2>&1
  cat $O/newton_cfg4.txt
  timeout 600 python scripts/newton_step_time.py cfg4 --re 10 100 --supg 0.05 > $O/newton_cfg4_supg.txt 2>&1
  cat $O/newton_cfg4_supg.txt
fi
if [ "$PART" = prof ]; then
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $GRAFT_REPO_ROOT/scripts/newton_step_time.py cfg4 --re 10 --supg 0.05 > $O/prof.out 2> $O/prof.err
  cd $GRAFT_REPO_ROOT
  find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_newton_supg.csv
  rm -rf $O/prof
  head -25 $O/kernel_stats_newton_supg.csv | cut -c1-200
fi
