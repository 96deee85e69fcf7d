"""Where the cycles of patch_invert_mfma_kernel go: per-wave phase sums of a -DALFI_INVERT_TIMING build (kernels_invert.hip).
usage: ALFI_HIP_LIB=<timing build> python scripts/invert_phases.py cfg4s"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from alfi_amd import hip, _lib

lv, tr, k = bench.build_problem(sys.argv[1], False)
L = lv[-1]
ctx = hip.Context(0)
dl = hip.Level(ctx, L.A, L.bc_dofs)
dl.set_patches(L.patch_ptr, L.patch_dofs)
dl.factor(); dl.factor(); ctx.sync()
lib = _lib.load()
fn = lib.alfi_debug_invert_phases
fn.restype = ctypes.c_int64
fn.argtypes = [ctypes.c_void_p]
out = np.zeros((4096, 8, 8), dtype=np.int64)
assert fn(out.ctypes.data) == 4096
npat = min(4096, len(L.patch_ptr) - 1)
o = out[:npat]
n = o[:, 0, 7]
sel = n == n.max()
o = o[sel]
o = o[:, o[0, :, 6] > 0]          # the waves of the build (8, or 4 with -DALFI_INVERT_WR=2)
steps = (n.max() + 3) // 4
names = ["panels -> LDS + barrier", "LU of the pivot block", "operands (substitutions)", "update MFMAs issued", "column fix-up", "row fix-up"]
print("%s: %d patches of %d dofs (%d block steps); shader-clock cycles per wave and block step, mean over waves / slowest wave"
      % (sys.argv[1], int(sel.sum()), int(n.max()), steps))
tot = o[:, :, 6].mean() / steps
for i, nm in enumerate(names):
    print("  %-28s %8.1f  (%4.1f %%)   max over waves %8.1f" % (nm, o[:, :, i].mean() / steps, 100 * o[:, :, i].mean() / steps / tot, o[:, :, i].mean(0).max() / steps))
print("  %-28s %8.1f" % ("whole kernel / steps", tot))
print("  per wave (mean over patches), phases x waves:")
print(np.round(o[:, :, :6].mean(0).T / steps, 0))
