"""Outer solve of one Newton step on the GPU (alfi_saddle_*; alfi/solver.py:386-422) against the oracle's restatement:
operator and block preconditioner applications, FGMRES iteration counts and solutions, and the reference's qualitative
promise -- outer iteration counts that stay small and flat under mesh refinement for moderate Reynolds numbers.  -m gpu"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from alfi_amd.problem import (TwoDimLidDrivenCavityProblem, ThreeDimLidDrivenCavityProblem, build_hierarchy,
                              build_pressure_coupling)


def _rhs(L, n_p, seed=0):
    f = np.random.default_rng(seed).standard_normal(L.n)
    f[L.bc_dofs] = 0.0
    return np.concatenate([f, np.zeros(n_p)])


@pytest.mark.parametrize("mk,ke,nref,Re,k", [(lambda: TwoDimLidDrivenCavityProblem(4), 2, 2, 100.0, 6),
                                             (lambda: ThreeDimLidDrivenCavityProblem(2), 2, 1, 100.0, 4),
                                             (lambda: ThreeDimLidDrivenCavityProblem(2), 1, 1, 10.0, 4)])
def test_outer_solve_matches_oracle(mk, ke, nref, Re, k):
    import alfi_amd
    from alfi_amd import hip
    from oracle import alfi_oracle as O
    prob = mk()
    lv, tr = build_hierarchy(prob, nref, ke, Re=Re)
    L = lv[-1]
    ctx = hip.Context(0)
    params = alfi_amd.outer_solver(prob.dim, alfi_amd.fieldsplit_0_mg(alfi_amd.mg_levels_solver(prob.dim, smoothing=k)))
    solver = alfi_amd.HipOuterSolver(ctx, lv, tr, params)
    B, vol = build_pressure_coupling(L)
    A = L.A.to_scipy().tocsr()
    omg = O.build_oracle_mg(lv, tr, k)
    b = _rhs(L, B.shape[0])
    n = L.n
    # operator and preconditioner applications
    v = np.random.default_rng(1).standard_normal(n + B.shape[0])
    v[L.bc_dofs] = 0.0
    dv, dy = ctx.vec(v), ctx.vec(n + B.shape[0])
    solver.saddle.mult(dv, dy)
    ref = np.concatenate([A @ v[:n] + B.T @ v[n:], B @ v[:n]])
    assert np.abs(dy.get() - ref).max() < 1e-12 * np.abs(ref).max()
    solver.saddle.precond(dv, dy)
    yu = omg.fcycle(v[:n])
    yp = -(L.nu + L.gamma) / vol * (v[n:] - B @ yu)
    yu = omg.fcycle(v[:n] - B.T @ yp)
    refp = np.concatenate([yu, yp - yp.mean()])
    got = dy.get()
    assert np.abs(got[:n] - refp[:n]).max() < 1e-5 * np.abs(refp[:n]).max()
    assert np.abs(got[n:] - refp[n:]).max() < 1e-5 * np.abs(refp[n:]).max()
    # the solve
    u, p, its, rn = solver.solve(b[:n])
    xo, its_o, hist = O.saddle_solve(omg, A, B, vol, L.nu, L.gamma, b, rtol=solver.rtol, atol=solver.atol)
    assert abs(its - its_o) <= 1, (its, its_o)
    assert rn <= 10 * max(solver.rtol * np.linalg.norm(b), solver.atol)
    assert np.abs(B @ u).max() < 1e-6 * np.abs(u).max()          # discretely divergence-free to solver tolerance
    assert np.abs(u - xo[:n]).max() < 1e-5 * np.abs(xo[:n]).max()
    po = xo[n:] - xo[n:].mean()
    assert np.abs((p - p.mean()) - po).max() < 1e-4 * np.abs(po).max()
    solver.saddle.close()
    solver.hmg.mg.close()
    ctx.close()


def test_outer_iterations_flat_under_refinement():
    """The point of the augmented-Lagrangian preconditioner (README.md:3; the papers' iteration tables): a handful of
    outer FGMRES iterations, not growing with the mesh, at Re 10 and 100 on ldc2d."""
    import alfi_amd
    from alfi_amd import hip
    ctx = hip.Context(0)
    counts = {}
    for Re in (10.0, 100.0):
        for nref in (2, 3, 4):
            prob = TwoDimLidDrivenCavityProblem(8)
            lv, tr = build_hierarchy(prob, nref, 2, Re=Re)
            params = alfi_amd.outer_solver(2, alfi_amd.fieldsplit_0_mg(alfi_amd.mg_levels_solver(2)))
            s = alfi_amd.HipOuterSolver(ctx, lv, tr, params)
            f = _rhs(lv[-1], 0, seed=2)
            u, p, its, rn = s.solve(f)
            counts[(Re, nref)] = its
            assert rn <= 10 * max(s.rtol * np.linalg.norm(f), s.atol)
            s.saddle.close()
            s.hmg.mg.close()
    assert max(counts.values()) <= 8, counts
    for Re in (10.0, 100.0):
        assert counts[(Re, 4)] <= counts[(Re, 2)] + 1, counts
    ctx.close()
