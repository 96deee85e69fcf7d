"""Scott-Vogelius side (towards BASELINE config 5): [P2]^d on Alfeld-split meshes with the full grad-div term, macro-star
patches, the macro-cell Schoeberl transfer (alfi/solver.py:604-662, alfi/transfer.py:49-88, 293-309, alfi/bary.py).
CPU: generator against the oracle's quadrature assembly, polynomial reproduction of the non-nested prolongation, and the
reference's gamma-robustness experiment (examples/graddiv/graddiv.py with --discretisation sv) on the oracle."""
import numpy as np
import pytest

from alfi_amd.problem import TwoDimLidDrivenCavityProblem
from alfi_amd.sv import build_sv_hierarchy
from oracle import alfi_oracle as O
from tests.test_graddiv import fgmres_solve


def test_sv_generator_against_quadrature_and_polynomials():
    lv, tr = build_sv_hierarchy(TwoDimLidDrivenCavityProblem(2), 2, 2, Re=0, gamma=100.0, advect=False)
    assert [L.n for L in lv] == [114, 418, 1602]
    assert all(np.diff(L.patch_ptr).max() == 62 for L in lv[1:])           # interior macro vertex, literal MacroStar
    assert tr[0].blk_dofs.shape == (8, 38) and tr[1].blk_dofs.shape == (32, 38)
    L, T = lv[1], tr[0]
    A = O.apply_bcs_matrix(O.assemble_form(L.V, nu=L.nu, gamma_full=100.0), L.bc_dofs)
    assert abs(A - L.A.to_scipy()).max() < 1e-11
    Ks, Ds = O.assemble_form(L.V, nu=1.0).tocsr(), O.assemble_form(L.V, gamma_full=1.0).tocsr()
    for b in (0, 5):
        d = T.blk_dofs[b]
        assert np.abs(Ks[d][:, d].toarray() - T.K_II[b]).max() < 1e-12
        assert np.abs(Ds[d][:, d].toarray() - T.D_II[b]).max() < 1e-12
    assert abs(Ds[T.blk_dofs.ravel()] - T.D_I.to_scipy()).max() < 1e-12
    # the blocks are the interiors of the coarse macro cells: disjoint, and together with the skeleton they cover everything
    assert len(np.unique(T.blk_dofs)) == T.blk_dofs.size
    assert len(np.unique(T.blk_dofs)) + len(T.skeleton_dofs) == L.n
    # non-nested prolongation reproduces P2 exactly
    def f(X):
        return np.stack([X[:, 0] ** 2 - X[:, 1], X[:, 0] * X[:, 1] + 1.0], axis=1).ravel()
    P = T.P.to_scipy()
    assert np.abs(P @ f(lv[0].V.node_coords) - f(L.V.node_coords)).max() < 1e-13
    # ... and inject on the non-nested levels (point evaluation of the fine field at the coarse nodes, solver.py:641-644)
    J = T.inject_matrix
    assert np.abs(J @ f(L.V.node_coords).reshape(-1, 2) - f(lv[0].V.node_coords).reshape(-1, 2)).max() < 1e-13
    assert np.abs(np.asarray(J.sum(axis=1)).ravel() - 1.0).max() < 1e-13       # a partition of unity per coarse node
    # robust prolongation keeps a divergence-free coarse field (discretely) divergence-free up to O(nu / gamma)
    ot = O.oracle_transfer(T, L, True).st
    uc = np.stack([lv[0].V.node_coords[:, 1] ** 2, lv[0].V.node_coords[:, 0] ** 2], axis=1).ravel()   # div = 0, in P2
    uf = ot.prolong(uc)
    assert uf @ (Ds @ uf) < 1e-6 * (uf @ (Ks @ uf))


def test_sv_p3_generator_3d():
    """[P3]^3 on Alfeld-split tetrahedra, the velocity space of BASELINE config 5 (solver.py:604-662 with k = 3)."""
    from alfi_amd.problem import ThreeDimLidDrivenCavityProblem
    lv, tr = build_sv_hierarchy(ThreeDimLidDrivenCavityProblem(1), 1, 3, Re=0, gamma=100.0, advect=False)
    assert [L.n for L in lv] == [462, 3189]
    L, T = lv[1], tr[0]
    assert np.diff(L.patch_ptr).max() == 1599 and T.blk_dofs.shape == (6, 390)
    A = O.apply_bcs_matrix(O.assemble_form(L.V, nu=L.nu, gamma_full=100.0), L.bc_dofs)
    assert abs(A - L.A.to_scipy()).max() < 1e-10
    Ks, Ds = O.assemble_form(L.V, nu=1.0).tocsr(), O.assemble_form(L.V, gamma_full=1.0).tocsr()
    d = T.blk_dofs[3]
    assert np.abs(Ks[d][:, d].toarray() - T.K_II[3]).max() < 1e-12
    assert np.abs(Ds[d][:, d].toarray() - T.D_II[3]).max() < 1e-12

    def f(X):
        return np.stack([X[:, 0] ** 3 - X[:, 1] * X[:, 2], X[:, 0] * X[:, 1] ** 2 + 1.0, X[:, 2] ** 3 - X[:, 0]], axis=1).ravel()
    P = T.P.to_scipy()
    assert np.abs(P @ f(lv[0].V.node_coords) - f(L.V.node_coords)).max() < 1e-12
    J = T.inject_matrix
    assert np.abs(J @ f(L.V.node_coords).reshape(-1, 3) - f(lv[0].V.node_coords).reshape(-1, 3)).max() < 1e-12
    ot = O.oracle_transfer(T, L, True).st
    X = lv[0].V.node_coords
    uc = np.stack([X[:, 1] ** 3, X[:, 2] ** 3, X[:, 0] ** 3], axis=1).ravel()             # div = 0, in P3
    uf = ot.prolong(uc)
    assert abs(uf @ (Ds @ uf)) < 1e-8 * (uf @ (Ks @ uf))


def test_bfs3d_problem_has_a_natural_outflow():
    """examples/bfs3d/bfs3d.py:23-26: Dirichlet data on the inflow and the walls only; the nodes strictly inside the outflow
    face x = 10 stay free, get matrix rows and sit in patches."""
    from alfi_amd.problem import ThreeDimBackwardsFacingStepProblem
    lv, tr = build_sv_hierarchy(ThreeDimBackwardsFacingStepProblem(1), 1, 2, Re=500.0, gamma=1e4)
    L = lv[1]
    X = L.V.node_coords
    out = np.abs(X[:, 0] - 10.0) < 1e-12
    wall = (np.abs(X[:, 1] - 2.0) < 1e-12) | (np.abs(X[:, 2]) < 1e-12) | (np.abs(X[:, 2] - 1.0) < 1e-12) | \
        (np.abs(X[:, 1]) < 1e-12) | (np.abs(X[:, 0]) < 1e-12) | \
        ((np.abs(X[:, 1] - 1.0) < 1e-12) & (X[:, 0] < 1 + 1e-12)) | ((np.abs(X[:, 0] - 1.0) < 1e-12) & (X[:, 1] < 1 + 1e-12))
    assert np.array_equal(L.V.bc_node_mask, wall)
    assert (out & ~wall).sum() > 0
    in_patch = np.zeros(L.n, dtype=bool)
    in_patch[L.patch_dofs] = True
    assert np.array_equal(in_patch.reshape(-1, 3)[:, 0], ~wall)
    assert L.nu == 1.0 / 500.0                                       # char_length = char_velocity = 1, problem.py:40-44
    A = O.apply_bcs_matrix(O.assemble_form(L.V, nu=L.nu, gamma_full=1e4, adv=1.0, wind=ThreeDimBackwardsFacingStepProblem(1).driver(X)),
                           L.bc_dofs)
    assert abs(A - L.A.to_scipy()).max() < 1e-8 * abs(A).max()


def run(gamma, schoeberl):
    lv, tr = build_sv_hierarchy(TwoDimLidDrivenCavityProblem(2), 2, 2, Re=0, gamma=gamma, advect=False)
    mg = O.build_oracle_mg(lv, tr, k=3, schoeberl_restriction=schoeberl)
    A = mg.levels[-1]["A"]
    b = np.ones(A.shape[0])
    b[lv[-1].bc_dofs] = 0
    return fgmres_solve(A, lambda r: mg.vcycle(len(lv) - 1, r, np.zeros_like(r)), b)


def test_sv_gamma_robustness():
    its = {g: run(g, True) for g in (0.0, 1e2, 1e4, 1e6)}
    assert max(its.values()) <= 12 and its[1e6] - its[1e2] <= 2, its


@pytest.mark.parametrize("dim,k", [(2, 2), (3, 3)])
def test_sv_pressure_coupling(dim, k):
    """Discontinuous P_{k-1} pressure of the Scott-Vogelius pair (solver.py:624-629): div [P_k]^d lies in it, so the full
    grad-div term is gamma B^T M^-1 B exactly, and M^-1 is the cell-wise inverse DGMassInv applies (solver.py:24)."""
    from alfi_amd.problem import ThreeDimLidDrivenCavityProblem
    from alfi_amd.sv import build_sv_pressure_coupling
    prob = TwoDimLidDrivenCavityProblem(2) if dim == 2 else ThreeDimLidDrivenCavityProblem(1)
    lv, tr = build_sv_hierarchy(prob, 1, k, Re=0, gamma=1.0, advect=False, patches=False)
    L = lv[1]
    B, M, Minv = build_sv_pressure_coupling(L, zero_bc_columns=False)
    npl = {1: dim + 1, 2: (dim + 1) * (dim + 2) // 2}[k - 1]
    assert B.shape == (L.V.mesh.num_cells * npl, L.n)
    assert abs((M @ Minv) - np.eye(M.shape[0])).max() < 1e-10
    D = O.assemble_form(L.V, gamma_full=1.0)
    assert abs(D - B.T @ Minv @ B).max() < 1e-10 * abs(D).max()
    # constants: B^T 1 = -int div(phi) = boundary flux only -> zero on interior velocity dofs
    flux = B.T @ np.ones(B.shape[0])
    interior = np.setdiff1d(np.arange(L.n), L.bc_dofs)
    assert abs(flux[interior]).max() < 1e-12


def test_sv_outer_solve_on_the_oracle():
    """One Newton-step linear solve of the Scott-Vogelius discretisation (2-D, [P2]^2-P1dg, Re 10): FGMRES + fieldsplit
    Schur full with the block DGMassInv converges in a handful of iterations (the reference's point, README.md:3)."""
    from alfi_amd.sv import build_sv_pressure_coupling
    lv, tr = build_sv_hierarchy(TwoDimLidDrivenCavityProblem(2), 2, 2, Re=10.0, gamma=1e4)
    L = lv[-1]
    B, M, Minv = build_sv_pressure_coupling(L)
    mg = O.build_oracle_mg(lv, tr, 6, schoeberl_restriction=False)            # 6 smoothing steps in 2-D, solver.py:309
    rng = np.random.default_rng(1)
    b = rng.standard_normal(L.n)
    b[L.bc_dofs] = 0
    rhs = np.concatenate([b, np.zeros(B.shape[0])])
    A = mg.levels[-1]["A"]
    x, its, hist = O.saddle_solve(mg, A, B, None, L.nu, L.gamma, rhs, rtol=1e-9, atol=1e-12, mass_inv=Minv)
    assert its <= 8, (its, hist)
    K = sp_bmat(A, B)
    assert np.linalg.norm(K @ x - rhs) <= 2e-9 * np.linalg.norm(rhs)


def sp_bmat(A, B):
    import scipy.sparse as sp
    return sp.bmat([[A, B.T], [B, None]], format="csr")


def test_macro_cell_groups_give_an_exact_block_factorisation():
    """The algebra behind the condensed patch factors (alfi_patches_set_groups), in NumPy on real macro-star patches: the dofs
    labelled with a macro cell are coupled to the rest of their patch only through unlabelled (skeleton) dofs, and
        t_g = X_g x_g,  y_S = inv(A_SS - sum_g B_g W_g)(x_S - sum_g B_g t_g),  y_g = t_g - W_g y_S
    reproduces inv(A_p) x.  Also the storage count sum_g (m_g^2 + 2 m_g s_g) + s^2 against n_p^2."""
    from alfi_amd.problem import ThreeDimLidDrivenCavityProblem
    from alfi_amd.sv import build_sv_hierarchy
    lv, _ = build_sv_hierarchy(ThreeDimLidDrivenCavityProblem(1), 1, 2, Re=100.0)
    L = lv[-1]
    A = L.A.to_scipy().tocsr()
    rng = np.random.default_rng(0)
    dense_total = cond_total = 0
    for p in range(len(L.patch_ptr) - 1):
        sl = slice(L.patch_ptr[p], L.patch_ptr[p + 1])
        dofs, lab = L.patch_dofs[sl], L.patch_groups[sl]
        Ap = A[dofs][:, dofs].toarray()
        n = len(dofs)
        S = np.flatnonzero(lab < 0)
        groups = [np.flatnonzero(lab == g) for g in np.unique(lab[lab >= 0])]
        assert len(groups) > 0 and len(S) > 0
        for i, gi in enumerate(groups):                     # no operator entry between two groups
            for gj in groups[i + 1:]:
                assert not Ap[np.ix_(gi, gj)].any() and not Ap[np.ix_(gj, gi)].any()
        x = rng.standard_normal(n)
        Sigma = Ap[np.ix_(S, S)].copy()
        rhs = x[S].copy()
        t, W = {}, {}
        cond = len(S) ** 2
        for k, g in enumerate(groups):
            Xg = np.linalg.inv(Ap[np.ix_(g, g)])
            Bg, Ags = Ap[np.ix_(S, g)], Ap[np.ix_(g, S)]
            W[k] = Xg @ Ags
            t[k] = Xg @ x[g]
            Sigma -= Bg @ W[k]
            rhs -= Bg @ t[k]
            sg = np.count_nonzero(np.abs(Ags).sum(axis=0) + np.abs(Bg).sum(axis=1))
            cond += len(g) ** 2 + 2 * len(g) * sg
        y = np.zeros(n)
        y[S] = np.linalg.solve(Sigma, rhs)
        for k, g in enumerate(groups):
            y[g] = t[k] - W[k] @ y[S]
        ref = np.linalg.solve(Ap, x)
        assert np.abs(y - ref).max() < 1e-8 * np.abs(ref).max()
        dense_total += n * n
        cond_total += cond
    assert cond_total < 0.35 * dense_total                  # [P2]^3 macro stars: 5.9 x on the device (whole-node groups)


@pytest.mark.parametrize("case", ["2d-P2", "3d-P2", "3d-P3", "bfs3d-P3"])
def test_vectorised_macro_stars_equal_the_literal_constructor(case):
    """sv.macro_star_patches_fast (sparse incidence products) against sv.macro_star_patches (the reference's MacroStar
    callback run point by point, relaxation.py:163-177): identical patch pointers, dofs and seed vertices on every level."""
    from alfi_amd import sv
    from alfi_amd.problem import (TwoDimLidDrivenCavityProblem, ThreeDimLidDrivenCavityProblem,
                                  ThreeDimBackwardsFacingStepProblem)
    prob, nref, k = {"2d-P2": (TwoDimLidDrivenCavityProblem(3), 1, 2), "3d-P2": (ThreeDimLidDrivenCavityProblem(2), 1, 2),
                     "3d-P3": (ThreeDimLidDrivenCavityProblem(1), 1, 3),
                     "bfs3d-P3": (ThreeDimBackwardsFacingStepProblem(1), 1, 3)}[case]
    lv, _ = sv.build_sv_hierarchy(prob, nref, k, Re=100.0)
    for L in lv[1:]:
        a, b = sv.macro_star_patches(L.V), sv.macro_star_patches_fast(L.V)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
        assert np.array_equal(L.patch_ptr, b[0]) and np.array_equal(L.patch_dofs, b[1])      # the hierarchy uses the fast one
