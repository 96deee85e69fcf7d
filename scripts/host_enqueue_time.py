"""Is a cycle of a small configuration bound by the host's launch rate?  Time to ENQUEUE one V-cycle (the call returns,
nothing waited for) against the time until the device has finished it.  usage: python scripts/host_enqueue_time.py cfg2"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                              # noqa: E402
from alfi_amd import hip                                  # noqa: E402

cfg = sys.argv[1]
lv, tr, k = bench.build_problem(cfg, False)
ctx = hip.Context(0)
mg = hip.Multigrid(ctx, lv, tr, k)
L = lv[-1]
b = np.random.default_rng(0).standard_normal(L.n)
b[L.bc_dofs] = 0.0
db, dx = ctx.vec(b), ctx.vec(L.n)
for _ in range(5):
    mg.vcycle(db, dx)
ctx.sync()
enq, tot = [], []
for _ in range(20):
    t0 = time.perf_counter()
    mg.vcycle(db, dx)
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    enq.append(t1 - t0)
    tot.append(t2 - t0)
print("%s: enqueue %.3f ms, until done %.3f ms (medians of 20 single cycles)" % (cfg, 1e3 * np.median(enq), 1e3 * np.median(tot)))
# back to back: 20 cycles enqueued, one sync
t0 = time.perf_counter()
for _ in range(20):
    mg.vcycle(db, dx)
t1 = time.perf_counter()
ctx.sync()
t2 = time.perf_counter()
print("%s: 20 cycles back to back: enqueue %.3f ms per cycle, done %.3f ms per cycle" % (cfg, 1e3 * (t1 - t0) / 20, 1e3 * (t2 - t0) / 20))
