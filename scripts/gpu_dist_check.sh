#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q > gpurun_out/pytest_gpu_dist.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu_dist.log
tail -40 gpurun_out/pytest_gpu_dist.log
ALFI_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --config cfg4s --verbose > gpurun_out/bench_dist2_gloo.log 2>&1
tail -15 gpurun_out/bench_dist2_gloo.log
