#!/bin/bash
# config 5 on one GPU: parity test at the small size, bench at nref 1 and 2, kernel trace of the nref-2 run
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_sv.py -m gpu -x -q -k "p3" > gpurun_out/pytest_cfg5.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_cfg5.log
tail -5 gpurun_out/pytest_cfg5.log
python bench.py --config cfg5s --steps 5 --warmup 2 > gpurun_out/bench_cfg5s.json 2> gpurun_out/bench_cfg5s.err
echo "bench cfg5s exit $?"; tail -3 gpurun_out/bench_cfg5s.err; cat gpurun_out/bench_cfg5s.json
python bench.py --config cfg5 --steps 5 --warmup 2 > gpurun_out/bench_cfg5.json 2> gpurun_out/bench_cfg5.err
echo "bench cfg5 exit $?"; tail -3 gpurun_out/bench_cfg5.err; cat gpurun_out/bench_cfg5.json
