"""Finest-level residual (one flat SpMV, mode 1) of a bench configuration, timed 40 x; env switches select kernel variants.
usage: python scripts/ab_spmv.py cfg4|cfg6"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                              # noqa: E402
from alfi_amd import hip                                  # noqa: E402
from alfi_amd.problem import ThreeDimLidDrivenCavityProblem, TwoDimLidDrivenCavityProblem, build_hierarchy   # noqa: E402

cfg = sys.argv[1]
dim, baseN, nref, ke, Re, k = bench.CONFIGS[cfg]
prob = TwoDimLidDrivenCavityProblem(baseN) if dim == 2 else ThreeDimLidDrivenCavityProblem(baseN)
lv, _ = build_hierarchy(prob, nref, ke, Re=Re, patches=False)
L = lv[-1]
ctx = hip.Context(0)
dl = hip.Level(ctx, L.A, L.bc_dofs)
rng = np.random.default_rng(0)
b, x, r = ctx.vec(rng.standard_normal(L.n)), ctx.vec(rng.standard_normal(L.n)), ctx.vec(L.n)
for _ in range(5):
    dl.residual(b, x, r)
ctx.sync()
t0 = time.time()
for _ in range(40):
    dl.residual(b, x, r)
ctx.sync()
t = (time.time() - t0) / 40
bytes_ = (8.0 * L.bs * L.bs + 4.0) * L.A.nnzb + 4.0 * (L.A.nbrows + 1) + 24.0 * L.n
print("%s XVEC=%s NT=%s: %.1f us, %.0f GB/s (%d dofs, %.1f blocks per row)" % (
    cfg, os.environ.get("ALFI_SPMV_XVEC", "1"), os.environ.get("ALFI_NT", "1"), t * 1e6, bytes_ / t / 1e9, L.n,
    L.A.nnzb / L.A.nbrows), flush=True)
