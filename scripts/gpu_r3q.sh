#!/bin/bash
# config 5 (Scott-Vogelius, condensed macro-star factors) over 4 mock ranks on one GPU: a functional record of the partitioned SV path
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3q
mkdir -p $O
MOCK=$(python -c "from tests.mock_rccl.build import build; print(build())")
ALFI_DIST_BACKEND=gloo ALFI_DIST_TRANSPORT=rccl ALFI_RCCL_LIB=$MOCK ALFI_DIST_MIN_DOFS=50000 python bench.py --gpus 4 --config cfg5 --steps 2 --warmup 1 > $O/r03_bench_dist4_cfg5_native_transport_mock_sharedgpu_functional.json 2> $O/bench_dist4_cfg5.err
tail -4 $O/bench_dist4_cfg5.err
head -c 700 $O/r03_bench_dist4_cfg5_native_transport_mock_sharedgpu_functional.json; echo
ALFI_DIST_BACKEND=gloo ALFI_DIST_TRANSPORT=rccl ALFI_RCCL_LIB=$MOCK ALFI_DIST_MIN_DOFS=400000 python bench.py --gpus 4 --config cfg5L --steps 2 --warmup 1 > $O/r03_bench_dist4_cfg5L_native_transport_mock_sharedgpu_functional.json 2> $O/bench_dist4_cfg5L.err
tail -4 $O/bench_dist4_cfg5L.err
head -c 700 $O/r03_bench_dist4_cfg5L_native_transport_mock_sharedgpu_functional.json; echo
