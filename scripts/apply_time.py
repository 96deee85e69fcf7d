"""Time of the additive patch apply (stage 1 + dof-wise sum) on every smoothed level of a bench configuration.
usage: python scripts/apply_time.py cfg3 [repetitions]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from alfi_amd import hip

lv, tr, k = bench.build_problem(sys.argv[1], False)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ctx = hip.Context(0)
for li in range(len(lv) - 1, 0, -1):
    L = lv[li]
    dl = hip.Level(ctx, L.A, L.bc_dofs)
    dl.set_patches(L.patch_ptr, L.patch_dofs)
    dl.factor()
    x = np.random.default_rng(5).standard_normal(L.n)
    dx, dy = ctx.vec(x), ctx.vec(L.n)
    for _ in range(5):
        dl.patch_apply(dx, dy)
    ctx.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.time()
        for _ in range(reps):
            dl.patch_apply(dx, dy)
        ctx.sync()
        best = min(best, (time.time() - t0) / reps)
    n2 = int((np.diff(L.patch_ptr).astype(np.int64) ** 2).sum())
    print("%s level %d: %d patches (max %d dofs), apply + sum %.1f us, %.2f TB/s of inverse bytes"
          % (sys.argv[1], li, len(L.patch_ptr) - 1, int(np.diff(L.patch_ptr).max()), best * 1e6, 8 * n2 / best / 1e12))
    del dl
