#!/usr/bin/env python3
"""Per-level device-time table of one V-cycle (HIP events of the library's profiling classes): python scripts/level_report.py cfg2"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from alfi_amd import hip, _lib

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
reps = 5
lv, tr, k = bench.build_problem(cfg, False)
ctx = hip.Context(0)
mg = hip.Multigrid(ctx, lv, tr, k)
L = lv[-1]
b = np.random.default_rng(0).standard_normal(L.n); b[L.bc_dofs] = 0
db, dx = ctx.vec(b), ctx.vec(L.n)
for _ in range(2): mg.vcycle(db, dx)
ctx.prof_enable(True); ctx.prof_reset()
for _ in range(reps): mg.vcycle(db, dx)
ctx.sync()
print("%-6s %10s %8s " % ("level", "dofs", "patches") + " ".join("%13s" % e for e in _lib.EVENTS[:8]))
for Lh, dl in zip(lv, mg.levels):
    p = ctx.prof_get(dl.id)
    print("%-6d %10d %8d " % (Lh.level, Lh.n, dl.patch_stats()[0]) + " ".join("%7.3f/%-5d" % (p[e][0] / reps, p[e][1] // reps) for e in _lib.EVENTS[:8]))
