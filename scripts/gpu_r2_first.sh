#!/bin/bash
# round 2, first pass: box facts, partitioned tests with both transports, self-spawning bench, forced 1-rank exchange overhead
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
  echo "== box"; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/memory.max 2>/dev/null; free -g | head -2
  rocm-smi --showmeminfo vram 2>/dev/null | head -8
  python - <<'PY'
import torch
print("devices", torch.cuda.device_count())
PY
} > gpurun_out/box.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_dist.py -m gpu -x -q > gpurun_out/pytest_dist.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_dist.log
tail -15 gpurun_out/pytest_dist.log
# self-spawn: two ranks sharing the GPU over gloo (functional)
ALFI_DIST_BACKEND=gloo ALFI_DIST_MIN_DOFS=1000 timeout 600 python bench.py --gpus 2 --config tiny --steps 3 --warmup 1 > gpurun_out/spawn2.json 2> gpurun_out/spawn2.err
echo "spawn exit $?"; tail -3 gpurun_out/spawn2.err; cat gpurun_out/spawn2.json
# a failing rank must fail the launcher
ALFI_DIST_BACKEND=nccl timeout 300 python bench.py --gpus 2 --config tiny --steps 1 --warmup 0 > gpurun_out/spawn_fail.out 2>&1; echo "spawn with too few GPUs exit $? (expected non-zero)"
for cfg in cfg4t cfg4s; do
python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg plain       ms/cycle %.3f' % d['ms_per_step'])"
for tp in rccl callback; do
ALFI_DIST_TRANSPORT=$tp ALFI_DIST_FORCE=1 ALFI_DIST_MIN_DOFS=1000 ALFI_DIST_OVERLAP=0 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline 2>gpurun_out/forced_$cfg_$tp.err | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg forced transport=$tp ms/cycle %.3f' % d['ms_per_step'], d.get('events_ms_rank0'))"
done
done 2>&1 | tee gpurun_out/forced_overhead.txt
