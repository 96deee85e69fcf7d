"""Where the time of a persistent multiplicative sweep goes: per-item clock stamps of a -DALFI_MULT_TIMING build of the library
(kernels_patch.hip: ticket | tables | wait over | residual | apply | stores drained | successors released), summarised.
usage: ALFI_HIP_LIB=<timing build> python scripts/mult_stamps.py cfg4"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from alfi_amd import hip, _lib
from alfi_amd.relaxation import OrderedRelaxation, Options

lv, tr, k = bench.build_problem(sys.argv[1], False)
L = lv[-1]
ctx = hip.Context(0)
dl = hip.Level(ctx, L.A, L.bc_dofs)
dl.set_patches(L.patch_ptr, L.patch_dofs)
dl.factor()
orl = OrderedRelaxation()
orl.name = "Star"
orl.opts = Options("", {"pc_patch_construction_Star_sort_order": "0+:1-"})
iterset = orl.iteration_order(L.V.mesh.coords[L.patch_seeds])
nw = dl.set_multiplicative(iterset, True)
x = np.random.default_rng(5).standard_normal(L.n)
x[L.bc_dofs] = 0.0
dx, dy = ctx.vec(x), ctx.vec(L.n)
for _ in range(3):
    dl.patch_apply(dx, dy)
ctx.sync()
lib = _lib.load()
fn = lib.alfi_debug_mult_stamps
fn.restype = ctypes.c_int64
fn.argtypes = [ctypes.c_void_p, ctypes.c_int64]
nitems = 2 * (len(L.patch_ptr) - 1)
st = np.zeros((nitems, 8), dtype=np.int64)
rc = fn(st.ctypes.data, nitems)
assert rc == nitems, rc
t = (st[:, :7] - st[:, 0].min()) * 0.01      # us (100 MHz)
names = ["tables (ticket -> tables read)", "wait (tables -> predecessors done)", "residual", "apply", "update + drain", "release"]
print("%s: %d items, %d wavefronts per sweep, whole sweep %.1f us" % (sys.argv[1], nitems, nw, t[:, 6].max() - t[:, 0].min()))
for i, nm in enumerate(names):
    d = t[:, i + 1] - t[:, i]
    print("  %-40s mean %7.2f  median %7.2f  p90 %7.2f  max %8.2f us" % (nm, d.mean(), np.median(d), np.percentile(d, 90), d.max()))
busy = t[:, 6] - t[:, 2]
print("  %-40s mean %7.2f  median %7.2f  p90 %7.2f us" % ("wait over -> released (the chain link)", busy.mean(), np.median(busy), np.percentile(busy, 90)))
print("  lower bound of the sweep from the chain: %d wavefronts x median link = %.1f us" % (2 * nw, 2 * nw * np.median(busy)))
# items in flight over time
ev = np.concatenate([np.stack([t[:, 2], np.ones(nitems)], 1), np.stack([t[:, 6], -np.ones(nitems)], 1)])
ev = ev[np.argsort(ev[:, 0])]
infl = np.cumsum(ev[:, 1])
dt = np.diff(ev[:, 0])
print("  items past their wait at a time: time-average %.1f" % ((infl[:-1] * dt).sum() / dt.sum()))
