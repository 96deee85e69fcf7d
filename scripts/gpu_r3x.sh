#!/bin/bash
# config 5 on the reference's gmsh channel (cfg5m: 440 k dofs; cfg5mL: 3.47 M dofs): full-size test, bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03f
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "cfg5m" 2>&1 | tail -5
python bench.py --config cfg5m --steps 10 --warmup 3 > $O/r03_bench_cfg5m.json 2> $O/bench_cfg5m.err
tail -3 $O/bench_cfg5m.err; head -c 700 $O/r03_bench_cfg5m.json; echo
timeout 1500 python bench.py --config cfg5mL --steps 5 --warmup 2 --no-cpu-baseline > $O/r03_bench_cfg5mL.json 2> $O/bench_cfg5mL.err
tail -3 $O/bench_cfg5mL.err; head -c 700 $O/r03_bench_cfg5mL.json; echo
