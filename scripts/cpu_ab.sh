#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
cat /sys/fs/cgroup/cpu.max; nproc
cat > /tmp/cpu_ab.py <<'PY'
import sys, os, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
import bench
from bench_cpu import cpu_baseline
t=time.time()
lv, tr, k = bench.build_problem("cfg4s", False)
print("gen", time.time()-t)
r = cpu_baseline("cfg4s", lv + [lv[-1]], tr + [tr[-1]], k, 1.0)
print(os.environ.get("OMP_SCHEDULE"), os.environ.get("OMP_NUM_THREADS"), r["cores"], r["sample"])
PY
for sched in "dynamic,16" "static"; do
OMP_SCHEDULE=$sched timeout 120 python /tmp/cpu_ab.py 2>/dev/null | tail -2
done
OMP_SCHEDULE="dynamic,16" OMP_NUM_THREADS=32 ALFI_HOST_THREADS=32 timeout 120 python /tmp/cpu_ab.py 2>/dev/null | tail -2
