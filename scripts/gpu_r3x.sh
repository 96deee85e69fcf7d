#!/bin/bash
# end of round 3: the Scott-Vogelius records (structured 5L, the reference's gmsh channel 5m / 5mL) and the config-4 Newton
# step times once more with the final code
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03f
mkdir -p $O
python bench.py --config cfg5m --steps 10 --warmup 3 > $O/r03_bench_cfg5m.json 2> $O/bench_cfg5m.err
head -c 260 $O/r03_bench_cfg5m.json; echo
python bench.py --config cfg5mL --steps 5 --warmup 2 --no-cpu-baseline > $O/r03_bench_cfg5mL.json 2> $O/bench_cfg5mL.err
head -c 260 $O/r03_bench_cfg5mL.json; echo
python bench.py --config cfg5L --steps 5 --warmup 2 --no-cpu-baseline > $O/r03_bench_cfg5L.json 2> $O/bench_cfg5L.err
head -c 260 $O/r03_bench_cfg5L.json; echo
python scripts/newton_step_time.py cfg4 --re 10 100 1000 > $O/r03_newton_cfg4_device_assembly.txt 2>&1
tail -n 4 $O/r03_newton_cfg4_device_assembly.txt
