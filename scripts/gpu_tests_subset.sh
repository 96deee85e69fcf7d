#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout ${TMO:-1500} python -m pytest $TESTS -m gpu -x -q -s > gpurun_out/pytest_subset.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_subset.log
tail -${TAILN:-40} gpurun_out/pytest_subset.log
