"""Patch factorisation time of the finest level of a bench configuration (gather + inversion + residual probe).
usage: python scripts/factor_time.py cfg4s"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from alfi_amd import hip
lv, tr, k = bench.build_problem(sys.argv[1], False)
L = lv[-1]
ctx = hip.Context(0)
dl = hip.Level(ctx, L.A, L.bc_dofs)
dl.set_patches(L.patch_ptr, L.patch_dofs)
dl.factor(); ctx.sync()
t0 = time.time()
for _ in range(3):
    dl.factor()
ctx.sync()
print("%s: factor %.1f ms per call (%d patches, max %d dofs; ALFI_INVERT_MFMA=%s)" % (sys.argv[1], (time.time() - t0) / 3 * 1e3, len(L.patch_ptr) - 1, np.diff(L.patch_ptr).max(), os.environ.get("ALFI_INVERT_MFMA", "default")))
print("patch check (worst residual, flagged, repaired, worst after):", dl.patch_check())
if len(sys.argv) > 2:      # y = M^-1 x for a fixed x: compared bitwise between kernel variants
    x = np.random.default_rng(5).standard_normal(L.n)
    dx, dy = ctx.vec(x), ctx.vec(L.n)
    dl.patch_apply(dx, dy)
    np.save(sys.argv[2], dy.get())
