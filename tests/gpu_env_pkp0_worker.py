"""Worker of tests/test_gpu_env_variants.py::test_smoother_paths: the FGMRES smoother on every level and the V- / F-cycle of a
2-D [P2]^2 and a 3-D [P2+FB]^3 hierarchy against the oracle, under whatever ALFI_* switches the parent put in the environment
(they select between the four-launch fused iteration and the general launch chain, and
between the direct and the de-duplicated x gathers of the large SpMV).  Prints ``SMOOTH <relerr>`` and ``CYCLE <relerr>``."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from alfi_amd import hip
    from alfi_amd.problem import (TwoDimLidDrivenCavityProblem, ThreeDimLidDrivenCavityProblem, build_hierarchy)
    from oracle import alfi_oracle as O
    ctx = hip.Context(0)
    worst_s = worst_c = 0.0
    for prob, ke, nref, Re, k in ((TwoDimLidDrivenCavityProblem(4), 2, 3, 100.0, 6),
                                  (ThreeDimLidDrivenCavityProblem(2), 2, 1, 1000.0, 4)):
        lv, tr = build_hierarchy(prob, nref, ke, Re=Re)
        mg = hip.Multigrid(ctx, lv, tr, k, robust_restriction=True)
        omg = O.build_oracle_mg(lv, tr, k, schoeberl_restriction=True)
        rng = np.random.default_rng(3)
        for L, dl, ol in list(zip(lv, mg.levels, omg.levels))[1:]:
            b, x0 = rng.standard_normal(L.n), rng.standard_normal(L.n)
            b[L.bc_dofs] = 0.0
            x0[L.bc_dofs] = 0.0
            for nonzero in (True, False):
                dx, db = ctx.vec(x0), ctx.vec(b)
                dl.smooth(k, db, dx, nonzero_guess=nonzero)
                ref = O.fgmres(lambda v: ol["A"] @ v, ol["smoother"].apply, b, x0 if nonzero else np.zeros_like(x0), k,
                               nonzero_guess=nonzero)
                worst_s = max(worst_s, np.abs(dx.get() - ref).max() / np.abs(ref).max())
            # zero right-hand side, zero guess: the iterate stays zero (no NaN from 0 / 0)
            dx, db = ctx.vec(L.n), ctx.vec(L.n)
            dl.smooth(k, db, dx, nonzero_guess=True)
            assert np.array_equal(dx.get(), np.zeros(L.n))
        L = lv[-1]
        b = rng.standard_normal(L.n)
        b[L.bc_dofs] = 0.0
        db, dx = ctx.vec(b), ctx.vec(L.n)
        top = len(lv) - 1
        mg.vcycle(db, dx)
        mg.vcycle(db, dx)
        ref = omg.vcycle(top, b, omg.vcycle(top, b, np.zeros(L.n)))
        worst_c = max(worst_c, np.abs(dx.get() - ref).max() / np.abs(ref).max())
        mg.fcycle(db, dx)
        ref = omg.fcycle(b)
        worst_c = max(worst_c, np.abs(dx.get() - ref).max() / np.abs(ref).max())
        mg.close()
    print("SMOOTH %.3e" % worst_s)
    print("CYCLE %.3e" % worst_c, flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
