#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3h
mkdir -p $O
START=$(date +%s)
timeout 2400 python -m pytest tests -x -q -m gpu --durations=25 > $O/pytest_all.log 2>&1
echo "pytest exit $? in $(( $(date +%s) - START )) s" >> $O/pytest_all.log
tail -40 $O/pytest_all.log
python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
