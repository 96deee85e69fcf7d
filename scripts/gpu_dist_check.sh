#!/bin/bash
# Functional check of the multi-rank leg of bench.py on the one GPU of the box: ranks share the device, halos staged
# through the host (gloo).  Never a performance number.
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
ALFI_DIST_BACKEND=gloo timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 4 --steps 2 --warmup 1 --outer > gpurun_out/bench_dist4_gloo.json 2> gpurun_out/bench_dist4_gloo.err
echo "exit $?"
tail -5 gpurun_out/bench_dist4_gloo.err
cat gpurun_out/bench_dist4_gloo.json
