"""The reference authors' own acceptance experiment for exactly this hot path (examples/graddiv/graddiv.py:38-180):
solve 2 sym grad u : grad v + gamma div u div v = (f, v) with multigrid-preconditioned Krylov for growing gamma; with
patch smoother + Schoeberl transfer the iteration count stays bounded, with plain transfer it blows up (the ">200"
sentinel of graddiv.py:161).  Here on the oracle (CPU); the GPU version is tests/test_gpu_graddiv.py."""
import numpy as np
import pytest

from alfi_amd.problem import TwoDimLidDrivenCavityProblem, ThreeDimLidDrivenCavityProblem, build_hierarchy
from oracle import alfi_oracle as O


def fgmres_solve(A, M, b, rtol=1e-8, maxit=60):
    V, Z, H = [b / np.linalg.norm(b)], [], np.zeros((maxit + 1, maxit))
    beta = np.linalg.norm(b)
    for j in range(maxit):
        z = M(V[j])
        Z.append(z)
        w = A @ z
        for i in range(j + 1):
            H[i, j] = V[i] @ w
            w = w - H[i, j] * V[i]
        H[j + 1, j] = np.linalg.norm(w)
        V.append(w / H[j + 1, j])
        e = np.zeros(j + 2)
        e[0] = beta
        y = np.linalg.lstsq(H[:j + 2, :j + 1], e, rcond=None)[0]
        if np.linalg.norm(H[:j + 2, :j + 1] @ y - e) < rtol * beta:
            return j + 1
    return maxit + 1


def run(prob, k, nref, gamma, schoeberl, ksmooth=3):
    lv, tr = build_hierarchy(prob, nref, k, Re=0, gamma=gamma, advect=False)
    mg = O.build_oracle_mg(lv, tr, k=ksmooth, schoeberl_restriction=schoeberl)
    if not schoeberl:
        for t in mg.transfers:                       # plain nodal/bubble transfer both ways (graddiv.py --transfer off)
            P = t.st.P
            t.prolong = (lambda P: (lambda xc: P @ xc))(P)
            t.restrict = (lambda P: (lambda rf: P.T @ rf))(P)
    A = mg.levels[-1]["A"]
    b = np.ones(A.shape[0])
    b[lv[-1].bc_dofs] = 0
    return fgmres_solve(A, lambda r: mg.vcycle(len(lv) - 1, r, np.zeros_like(r)), b)


@pytest.mark.parametrize("mk,k,nref", [(lambda: TwoDimLidDrivenCavityProblem(4), 2, 2),
                                       (lambda: ThreeDimLidDrivenCavityProblem(2), 1, 1)])
def test_gamma_robustness(mk, k, nref):
    its = {g: run(mk(), k, nref, g, True) for g in (0.0, 1e2, 1e4, 1e6)}
    assert max(its.values()) <= 12, its                     # bounded in gamma
    assert max(its.values()) - min(its.values()) <= 5, its
    assert run(mk(), k, nref, 1e6, False) > 40              # plain transfer: not robust
