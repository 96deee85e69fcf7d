"""Worker of tests/test_gpu_env_variants.py: one V-cycle of the 3-D [P3]^3 Scott-Vogelius hierarchy (large patches, large
transfer blocks, BSR SpMV, FGMRES) against the oracle, under whatever ALFI_* switches the parent put in the environment.
Prints ``RELERR <value>``."""
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
    lv, tr = build_sv_hierarchy(ThreeDimLidDrivenCavityProblem(1), 1, 3, Re=100.0, gamma=1e4)
    ctx = hip.Context(0)
    mg = hip.Multigrid(ctx, lv, tr, 3, robust_restriction=True)
    omg = O.build_oracle_mg(lv, tr, 3, schoeberl_restriction=True)
    L = lv[-1]
    b = np.random.default_rng(0).standard_normal(L.n)
    b[L.bc_dofs] = 0
    db, dx = ctx.vec(b), ctx.vec(L.n)
    mg.vcycle(db, dx)
    ref = omg.vcycle(1, b, np.zeros(L.n))
    print("RELERR %.3e" % (np.abs(dx.get() - ref).max() / np.abs(ref).max()), flush=True)
    mg.close()
    ctx.close()


if __name__ == "__main__":
    main()
