"""Time of one symmetrised multiplicative patch sweep (alfi_patch_apply with patch_pc_patch_local_type multiplicative,
symmetrise_sweep; alfi/solver.py:322-335) on the finest level of a bench configuration, and a checksum of its result.
usage: python scripts/mult_time.py cfg4 [out.npy]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from alfi_amd import hip
from alfi_amd.relaxation import OrderedRelaxation, Options

lv, tr, k = bench.build_problem(sys.argv[1], False)
L = lv[-1]
ctx = hip.Context(0)
dl = hip.Level(ctx, L.A, L.bc_dofs)
dl.set_patches(L.patch_ptr, L.patch_dofs)
dl.factor()
orl = OrderedRelaxation()
orl.name = "Star"
orl.opts = Options("", {"pc_patch_construction_Star_sort_order": "0+:1-"})   # ldc relaxation_direction (ldc2d.py:39)
iterset = orl.iteration_order(L.V.mesh.coords[L.patch_seeds])
nw = dl.set_multiplicative(iterset, True)
x = np.random.default_rng(5).standard_normal(L.n)
x[L.bc_dofs] = 0.0
dx, dy = ctx.vec(x), ctx.vec(L.n)
dl.patch_apply(dx, dy)
ctx.sync()
t0 = time.time()
for _ in range(5):
    dl.patch_apply(dx, dy)
ctx.sync()
ms = (time.time() - t0) / 5 * 1e3
n2 = int((np.diff(L.patch_ptr).astype(np.int64) ** 2).sum())
print("%s: symmetrised multiplicative apply %.2f ms (%d wavefronts per sweep, %d patches, %.2f TB/s counting the inverses twice)"
      % (sys.argv[1], ms, nw, len(L.patch_ptr) - 1, 2 * 8 * n2 / (ms * 1e-3) / 1e12))
y = dy.get()
print("checksum %.17g %.17g" % (float(np.abs(y).sum()), float(y @ x)))
if len(sys.argv) > 2:
    np.save(sys.argv[2], y)
