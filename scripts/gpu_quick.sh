#!/bin/bash
# quick GPU pass: parity tests + bench (no profiling)
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
python bench.py --no-cpu-baseline $BENCH_ARGS > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err
tail -3 gpurun_out/bench_quick.err
cat gpurun_out/bench_quick.json
