"""The reference's solve loop (Newton + Reynolds continuation, alfi/solver.py:257-300, alfi/driver.py:95-128) with every
linear solve on the GPU, against a host Newton iteration that solves the same discrete equations with a sparse direct
solver.  -m gpu"""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

pytestmark = pytest.mark.gpu

from alfi_amd.problem import TwoDimLidDrivenCavityProblem, ThreeDimLidDrivenCavityProblem, BSR


def host_newton(solver, re, u, p, tol):
    """Direct-solver Newton on the same residual / Jacobian (mean pressure fixed by a Lagrange multiplier)."""
    from alfi_amd.nssolver import _assemble
    L = solver.levels[-1]
    nu = solver.char_L * solver.char_U / re
    solver.nu = nu
    n, m = solver.n_u, solver.n_p
    vol = solver.vol
    for it in range(25):
        Fu, Fp = solver.residual(u, p, 1.0)
        if np.sqrt(Fu @ Fu + Fp @ Fp) < tol:
            break
        J = BSR(L.A.nbrows, L.A.nbcols, L.bs, L.A.rowptr, L.A.colidx,
                solver.level_values(L, u.reshape(-1, L.bs), 1.0, True)).to_scipy().tocsr()
        K = sp.bmat([[J, solver.B.T, None], [solver.B, None, sp.csr_matrix(vol[:, None])],
                     [None, sp.csr_matrix(vol[None, :]), None]], format="csc")
        d = spla.spsolve(K, -np.concatenate([Fu, Fp, [0.0]]))
        u = u + d[:n]
        p = p + d[n:n + m]
    return u, p - (vol @ p) / vol.sum(), it


@pytest.mark.parametrize("mk,ke,nref,disc", [(lambda: TwoDimLidDrivenCavityProblem(8), 2, 1, "pkp0"),
                                             (lambda: ThreeDimLidDrivenCavityProblem(2), 2, 1, "pkp0"),
                                             # Scott-Vogelius [P2]^2 - P1dg on the barycentric hierarchy, macro-star patches,
                                             # block DGMassInv, state moved to the coarse levels by point evaluation
                                             (lambda: TwoDimLidDrivenCavityProblem(4), 2, 1, "sv"),
                                             # SUPG-stabilised momentum equation (stabilisation.py:47-97; the authors'
                                             # production option, examples/generate_submission:18-20)
                                             (lambda: TwoDimLidDrivenCavityProblem(8), 2, 1, "supg"),
                                             (lambda: ThreeDimLidDrivenCavityProblem(2), 1, 1, "supg"),
                                             # the coarse grid re-factored at every Newton step by the multifrontal solver
                                             (lambda: ThreeDimLidDrivenCavityProblem(2), 2, 1, "pkp0-sparse-coarse")])
def test_newton_continuation_matches_direct_solver(mk, ke, nref, disc, monkeypatch):
    from alfi_amd.nssolver import HipNavierStokesSolver, run_solver
    prob = mk()
    if disc == "pkp0-sparse-coarse":
        monkeypatch.setenv("ALFI_COARSE_SPARSE_MIN", "0")
        disc = "pkp0"
    if disc == "supg":
        s = HipNavierStokesSolver(prob, nref, ke, stabilisation_type="supg", stabilisation_weight=0.05)
    else:
        s = HipNavierStokesSolver(prob, nref, ke, discretisation=disc)
    u0, p0 = s.u.copy(), s.p.copy()
    res = run_solver(s, [10, 100])
    for re in (10, 100):
        info = res[re]
        assert info["converged"], info
        assert info["nonlinear_iter"] <= 8
        # a handful of Krylov iterations per Newton step (the reference's raison d'etre, README.md:3)
        assert info["linear_iter"] <= 10 * info["nonlinear_iter"], info
        # Newton converges fast near the solution: the last step reduces the residual by > 100x
        h = info["residual_history"]
        assert h[-1] < 1e-2 * h[-2] or h[-1] < s.snes_atol        # (the last step is limited by the linear tolerance)
    u, p = s.u.copy(), s.p.copy()
    # host reference: same continuation, sparse direct solves
    uh, ph = u0, p0
    for re in (10, 100):
        uh, ph, _ = host_newton(s, re, uh, ph, 1e-10)
    assert np.abs(u - uh).max() < 1e-6 * np.abs(uh).max()
    assert np.abs(p - ph).max() < 1e-5 * max(np.abs(ph).max(), 1e-12)
    # discretely divergence-free, Dirichlet data intact
    assert np.abs(s.B_raw @ u).max() < 1e-7
    L = s.levels[-1]
    assert np.array_equal(u.reshape(-1, L.bs)[L.V.bc_nodes], prob.driver(L.V.node_coords[L.V.bc_nodes]))
    s.close()
