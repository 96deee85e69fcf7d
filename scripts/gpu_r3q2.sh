#!/bin/bash
# config 5 on the reference's gmsh channel over 4 mock ranks on one GPU, with one outer (Newton-step) solve on the partitioned
# levels: a functional record of the partitioned Scott-Vogelius path end to end
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3q
mkdir -p $O
MOCK=$(python -c "from tests.mock_rccl.build import build; print(build())")
ALFI_DIST_BACKEND=gloo ALFI_DIST_TRANSPORT=rccl ALFI_RCCL_LIB=$MOCK python bench.py --gpus 4 --config cfg5m --steps 2 --warmup 1 --outer > $O/r03_bench_dist4_cfg5m_outer_native_transport_mock_sharedgpu_functional.json 2> $O/bench_dist4_cfg5m.err
tail -4 $O/bench_dist4_cfg5m.err
head -c 600 $O/r03_bench_dist4_cfg5m_outer_native_transport_mock_sharedgpu_functional.json; echo
python bench.py --config cfg5m --steps 2 --warmup 1 --outer --no-cpu-baseline > $O/r03_bench_cfg5m_outer.json 2> $O/bench_cfg5m_outer.err
python - $O <<'PY'
import json, sys
a = json.load(open(sys.argv[1] + "/r03_bench_dist4_cfg5m_outer_native_transport_mock_sharedgpu_functional.json"))
b = json.load(open(sys.argv[1] + "/r03_bench_cfg5m_outer.json"))
print("4 ranks:", a.get("outer_solve") or a.get("outer"))
print("1 GPU  :", b.get("outer_solve"))
PY
