"""Worker of tests/test_gpu_condensed.py::test_flagged_condensed_factors_are_repaired_in_place: the parent sets
ALFI_PATCH_CHECK_TOL below what any inverse reaches, so EVERY condensed patch is flagged by the residual probe and its
Schur complement re-inverted by the pivoted LU (kernels_check.hip: cond_repair).  Prints
``CHECK <flagged> <repaired> <worst before> <worst after> BYTES <condensed> <dense> RELERR <apply vs oracle>``."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from alfi_amd import hip
    from alfi_amd.problem import ThreeDimLidDrivenCavityProblem
    from alfi_amd.sv import build_sv_hierarchy
    from oracle import alfi_oracle as O
    lv, _ = build_sv_hierarchy(ThreeDimLidDrivenCavityProblem(1), 1, 3, Re=100.0)
    L = lv[-1]
    ctx = hip.Context(0)
    dl = hip.Level(ctx, L.A, L.bc_dofs)
    dl.set_patches(L.patch_ptr, L.patch_dofs)
    dl.set_patch_groups(L.patch_groups)
    dl.factor()
    worst, flagged, repaired, after = dl.patch_check()
    x = np.random.default_rng(0).standard_normal(L.n)
    dx, dy = ctx.vec(x), ctx.vec(L.n)
    dl.patch_apply(dx, dy)
    ref = O.PatchSmoother(L.A.to_scipy().tocsr(), L.patch_ptr, L.patch_dofs, L.bc_dofs).apply(x)
    err = np.abs(dy.get() - ref).max() / np.abs(ref).max()
    dense = 8 * int((np.diff(L.patch_ptr).astype(np.int64) ** 2).sum())
    print("CHECK %d %d %.3e %.3e BYTES %d %d RELERR %.3e" % (flagged, repaired, worst, after, dl.factor_bytes(), dense, err),
          flush=True)
    dl.close()
    ctx.close()


if __name__ == "__main__":
    main()
