#!/bin/bash
# One GPU-box pass: parity tests, bench, kernel trace, PMC traffic passes.  Outputs under gpurun_out/.
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
python bench.py > gpurun_out/bench_cfg4.json 2> gpurun_out/bench_cfg4.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg4 -- python3 bench.py --no-cpu-baseline > gpurun_out/bench_cfg4_rocprof.json 2> gpurun_out/prof_cfg4.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 > gpurun_out/pmc_fetch.json 2> gpurun_out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 > gpurun_out/pmc_write.json 2> gpurun_out/pmc_write.err
ls -la gpurun_out gpurun_out/*/* | tail -40
tail -3 gpurun_out/pytest_gpu.log
cat gpurun_out/bench_cfg4.json
