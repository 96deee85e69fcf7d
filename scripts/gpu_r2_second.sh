#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 ./scripts/micro/gridbar > gpurun_out/gridbar.txt 2>&1; echo "gridbar exit $?"; cat gpurun_out/gridbar.txt
ALFI_DIST_BACKEND=gloo ALFI_DIST_MIN_DOFS=1000 timeout 600 python bench.py --gpus 3 --config cfg4t --steps 3 --warmup 1 > gpurun_out/spawn3.json 2> gpurun_out/spawn3.err
echo "spawn exit $?"; tail -3 gpurun_out/spawn3.err; cat gpurun_out/spawn3.json
for cfg in cfg4t; do
ALFI_BENCH_PROF=0 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg plain       ms/cycle %.3f' % d['ms_per_step'])"
for tp in rccl callback; do
ALFI_BENCH_PROF=0 ALFI_DIST_TRANSPORT=$tp ALFI_DIST_FORCE=1 ALFI_DIST_MIN_DOFS=1000 ALFI_DIST_OVERLAP=0 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline 2>gpurun_out/forced_$tp.err | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg forced transport=$tp ms/cycle %.3f' % d['ms_per_step'], d.get('events_ms_rank0'))"
done
done 2>&1 | tee gpurun_out/forced_overhead2.txt
for cfg in cfg2 cfg3; do
ALFI_BENCH_PROF=0 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null > gpurun_out/bench_$cfg.json
python -c "
import sys, json
d = json.load(open('gpurun_out/bench_$cfg.json')); print('$cfg ms/cycle %.3f' % d['ms_per_step'], d['events_ms'])"
done
