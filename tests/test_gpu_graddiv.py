"""The reference authors' acceptance experiment for this hot path (examples/graddiv/graddiv.py:38-180) on the GPU:
2 sym grad u : grad v + gamma cell_avg(div u) div v = (f, v), Krylov preconditioned by one device V-cycle (patch smoother
+ Schoeberl transfer); the iteration count stays bounded as gamma grows from 0 to 1e6 and matches the oracle's count.
-m gpu.  The CPU twin (oracle only, incl. the blow-up without the robust transfer) is tests/test_graddiv.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from alfi_amd.problem import TwoDimLidDrivenCavityProblem, ThreeDimLidDrivenCavityProblem, build_hierarchy
from tests.test_graddiv import fgmres_solve, run as run_oracle


def run_gpu(ctx, prob, k, nref, gamma, ksmooth=3):
    from alfi_amd import hip
    lv, tr = build_hierarchy(prob, nref, k, Re=0, gamma=gamma, advect=False)
    mg = hip.Multigrid(ctx, lv, tr, ksmooth, robust_restriction=True)
    L = lv[-1]
    A = L.A.to_scipy().tocsr()
    b = np.ones(L.n)
    b[L.bc_dofs] = 0
    dr, dz = ctx.vec(L.n), ctx.vec(L.n)

    def M(r):
        dr.set(r)
        dz.zero()
        mg.vcycle(dr, dz)
        return dz.get()
    its = fgmres_solve(A, M, b)
    mg.close()
    return its


@pytest.mark.parametrize("mk,k,nref", [(lambda: TwoDimLidDrivenCavityProblem(4), 2, 2),
                                       (lambda: ThreeDimLidDrivenCavityProblem(2), 1, 1),
                                       (lambda: ThreeDimLidDrivenCavityProblem(2), 2, 1)])
def test_gamma_robustness_on_the_gpu(mk, k, nref):
    from alfi_amd import hip
    ctx = hip.Context(0)
    its = {g: run_gpu(ctx, mk(), k, nref, g) for g in (0.0, 1e2, 1e4, 1e6)}
    assert max(its.values()) <= 12, its                     # bounded in gamma
    assert max(its.values()) - min(its.values()) <= 5, its
    ref = run_oracle(mk(), k, nref, 1e4, True)
    assert abs(its[1e4] - ref) <= 1, (its, ref)
    ctx.close()
