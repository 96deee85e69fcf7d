#!/bin/bash
# config 5 on one GPU: bench with the CPU baseline, kernel trace, MFMA-busy counters of the setup
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py --config cfg5 --steps 5 --warmup 2 > gpurun_out/bench_cfg5.json 2> gpurun_out/bench_cfg5.err
echo "bench cfg5 exit $?"; tail -3 gpurun_out/bench_cfg5.err; cat gpurun_out/bench_cfg5.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg5 -- python3 bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_cfg5_rocprof.json 2> gpurun_out/prof_cfg5.err
echo "rocprof exit $?"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_cfg5_mfma -- python3 bench.py --config cfg5s --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_cfg5_mfma.json 2> gpurun_out/pmc_cfg5_mfma.err
echo "pmc exit $?"
find gpurun_out/prof_cfg5 gpurun_out/pmc_cfg5_mfma -name "*.csv" | head
