"""Does the pivoted repair (kernels_check.hip) restore accuracy at full patch size?  Factor a level with the probe tolerance
set so low that (nearly) every patch is flagged and re-inverted with partial pivoting; report the residuals before / after.
usage: ALFI_PATCH_CHECK_TOL=1e-11 python scripts/repair_check.py cfg4s"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from alfi_amd import hip
lv, tr, k = bench.build_problem(sys.argv[1], False)
L = lv[-1]
ctx = hip.Context(0)
dl = hip.Level(ctx, L.A, L.bc_dofs)
dl.set_patches(L.patch_ptr, L.patch_dofs)
try:
    dl.factor()
    print("factor ok:", dl.patch_check())
except Exception as e:
    print("factor raised:", str(e)[:300])
