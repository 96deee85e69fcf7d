"""Sparse (multifrontal) coarse factorisation, alfi_coarse_factor_sparse: what the reference gets from MUMPS / SuperLU_DIST
(AssembledPC + LU, alfi/solver.py:369-378).  The oracle is SciPy's SuperLU on the same operator; the dense inverse of
alfi_coarse_factor is the second witness.  -m gpu."""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

from alfi_amd.problem import (ThreeDimLidDrivenCavityProblem, TwoDimLidDrivenCavityProblem, build_hierarchy)

pytestmark = pytest.mark.gpu


def _level(kind, N, nref=0, **kw):
    prob = TwoDimLidDrivenCavityProblem(N) if kind == "2d" else ThreeDimLidDrivenCavityProblem(N)
    lv, tr = build_hierarchy(prob, nref, kw.pop("k", 2 if kind == "2d" else 1), Re=kw.pop("Re", 10.0), **kw)
    return lv, tr


@pytest.mark.parametrize("kind,N,leaf,coords", [("2d", 8, 4, True), ("2d", 8, 4, False), ("2d", 16, 0, True),
                                                ("3d", 2, 6, True), ("3d", 3, 16, True), ("3d", 3, 16, False),
                                                ("3d", 4, 0, True)])
def test_sparse_factor_solves_like_superlu(kind, N, leaf, coords):
    from alfi_amd import hip
    lv, _ = _level(kind, N)
    L = lv[0]
    ctx = hip.Context(0)
    dl = hip.Level(ctx, L.A, L.bc_dofs)
    res = dl.coarse_factor_sparse(L.V.node_coords if coords else None, leaf_nodes=leaf)
    assert 0 <= res < 1e-7, res
    A = L.A.to_scipy().tocsc()
    lu = spla.splu(A)
    rng = np.random.default_rng(0)
    for _ in range(2):
        b = rng.standard_normal(L.n)
        want = lu.solve(b)
        bx, xx = ctx.vec(b), ctx.vec(L.n)
        dl.coarse_solve(bx, xx)
        got = xx.get()
        assert np.abs(got - want).max() < 1e-9 * np.abs(want).max(), np.abs(got - want).max() / np.abs(want).max()
    # the factors are smaller than the dense inverse once the tree has a few heights
    if L.n > 2000:
        assert dl.coarse_factor_bytes() < 8 * L.n * L.n
    # switching back to the dense inverse drops the sparse factors
    dl.coarse_factor()
    assert dl.coarse_factor_bytes() == 8 * L.n * L.n
    dl.coarse_solve(bx, xx)
    assert np.abs(xx.get() - want).max() < 1e-8 * np.abs(want).max()
    dl.close()
    ctx.close()


def test_cycle_with_sparse_coarse_solver_equals_dense():
    """A two-level hierarchy whose coarse solve goes through the multifrontal factors gives the V-cycle of the dense
    inverse (both are the exact solve to rounding)."""
    from alfi_amd import hip
    lv, tr = _level("3d", 3, nref=1)
    rng = np.random.default_rng(1)
    b = rng.standard_normal(lv[-1].n)
    b[lv[-1].bc_dofs] = 0.0
    out = {}
    for mode in ("dense", "sparse"):
        ctx = hip.Context(0)
        mg = hip.Multigrid(ctx, lv, tr, 6, coarse=mode)
        bx, xx = ctx.vec(b), ctx.vec(lv[-1].n)
        mg.vcycle(bx, xx)
        mg.vcycle(bx, xx)
        out[mode] = xx.get()
        mg.close()
        ctx.close()
    err = np.abs(out["dense"] - out["sparse"]).max() / np.abs(out["dense"]).max()
    assert err < 1e-9, err


def test_sparse_factor_of_a_refined_level_as_large_coarse_grid():
    """A coarse grid the size of a level-1 operator (3-D [P1+FB]^3, N = 8: 9k dofs), gamma = 1e4, Re = 1000 -- the regime in
    which the explicit front inverses lose digits and the per-solve refinement step matters."""
    from alfi_amd import hip
    lv, _ = _level("3d", 8, Re=1000.0)
    L = lv[0]
    ctx = hip.Context(0)
    dl = hip.Level(ctx, L.A, L.bc_dofs)
    res = dl.coarse_factor_sparse(L.V.node_coords)
    assert res < 1e-6, res
    lu = spla.splu(L.A.to_scipy().tocsc())
    b = np.random.default_rng(2).standard_normal(L.n)
    bx, xx = ctx.vec(b), ctx.vec(L.n)
    dl.coarse_solve(bx, xx)
    want = lu.solve(b)
    assert np.abs(xx.get() - want).max() < 1e-8 * np.abs(want).max()
    assert dl.coarse_factor_bytes() < 0.5 * 8 * L.n * L.n
    dl.close()
    ctx.close()


def test_single_front_and_disconnected_graph():
    """Edge cases of the ordering: an operator that fits one leaf (one front, no boundary, no sweeps beyond the root) and a
    block-diagonal operator of two uncoupled copies (the bisection finds halves that do not touch: a formal separator)."""
    import scipy.sparse as sp
    from alfi_amd import hip
    from alfi_amd.problem import BSR
    lv, _ = _level("2d", 4)
    L = lv[0]
    A = L.A.to_scipy().tocsr()
    ctx = hip.Context(0)
    rng = np.random.default_rng(3)
    # one leaf
    dl = hip.Level(ctx, L.A, L.bc_dofs)
    assert dl.coarse_factor_sparse(None, leaf_nodes=10 ** 6) < 1e-8
    b = rng.standard_normal(L.n)
    bx, xx = ctx.vec(b), ctx.vec(L.n)
    dl.coarse_solve(bx, xx)
    want = spla.splu(A.tocsc()).solve(b)
    assert np.abs(xx.get() - want).max() < 1e-9 * np.abs(want).max()
    dl.close()
    # two uncoupled copies, small leaves, with and without coordinates
    A2 = sp.block_diag([A, 2.0 * A]).tocsr()
    B2 = BSR.from_scipy(A2, L.A.bs)
    bc2 = np.concatenate([L.bc_dofs, L.bc_dofs + L.n]).astype(np.int32)
    xy = np.vstack([L.V.node_coords, L.V.node_coords + np.array([3.0, 0.0])])
    lu2 = spla.splu(A2.tocsc())
    for coords in (None, xy):
        d2 = hip.Level(ctx, B2, bc2)
        assert d2.coarse_factor_sparse(coords, leaf_nodes=5) < 1e-8
        b = rng.standard_normal(2 * L.n)
        bx, xx = ctx.vec(b), ctx.vec(2 * L.n)
        d2.coarse_solve(bx, xx)
        want = lu2.solve(b)
        assert np.abs(xx.get() - want).max() < 1e-9 * np.abs(want).max()
        d2.close()
    ctx.close()


def test_refactorisation_with_new_values_reuses_the_plan():
    """alfi_level_update_values + alfi_coarse_factor_sparse (what every Newton step does): the ordering and the symbolic
    factorisation are kept, the numeric factors follow the new values."""
    import time
    from alfi_amd import hip
    lv, _ = _level("3d", 4, k=2)
    L = lv[0]
    ctx = hip.Context(0)
    dl = hip.Level(ctx, L.A, L.bc_dofs)
    t0 = time.time()
    dl.coarse_factor_sparse(None)
    t_first = time.time() - t0
    A = L.A.to_scipy().tocsr()
    rng = np.random.default_rng(4)
    b = rng.standard_normal(L.n)
    bx, xx = ctx.vec(b), ctx.vec(L.n)
    # new values: another Reynolds number on the same sparsity
    lv2, _ = _level("3d", 4, k=2, Re=200.0)
    assert np.array_equal(lv2[0].A.colidx, L.A.colidx)
    dl.update_values(lv2[0].A.vals)
    t0 = time.time()
    assert dl.coarse_factor_sparse(None) < 1e-7
    t_again = time.time() - t0
    dl.coarse_solve(bx, xx)
    want = spla.splu(lv2[0].A.to_scipy().tocsc()).solve(b)
    assert np.abs(xx.get() - want).max() < 1e-9 * np.abs(want).max()
    old = spla.splu(A.tocsc()).solve(b)
    assert np.abs(xx.get() - old).max() > 1e-6 * np.abs(old).max()        # (it is the new operator that was factored)
    print("first factorisation %.3f s, with the plan %.3f s" % (t_first, t_again))
    dl.close()
    ctx.close()


def test_the_reference_largest_coarse_grid():
    """Full size: the coarse grid of the reference's largest defined run (ldc3d [P1+FB]^3, baseN 18,
    examples/generate_submission:10-23): 236 361 dofs -- 447 GB as a dense inverse.  Size-independent property instead of
    an oracle solve: the residual of the returned solution, computed on the host."""
    from alfi_amd import hip
    lv, _ = build_hierarchy(ThreeDimLidDrivenCavityProblem(18), 0, 1, Re=100.0, patches=False)
    L = lv[0]
    assert L.n == 236361
    ctx = hip.Context(0)
    dl = hip.Level(ctx, L.A, L.bc_dofs)
    res = dl.coarse_factor_sparse(None)
    assert res < 1e-7, res
    assert dl.coarse_factor_bytes() < 12e9          # (dense: 4.5e11)
    A = L.A.to_scipy().tocsr()
    rng = np.random.default_rng(5)
    for _ in range(2):
        b = rng.standard_normal(L.n)
        bx, xx = ctx.vec(b), ctx.vec(L.n)
        dl.coarse_solve(bx, xx)
        x = xx.get()
        r = np.abs(A @ x - b).max() / np.abs(b).max()
        assert r < 1e-8, r
    dl.close()
    ctx.close()
