"""The residual probe of every stored patch inverse and the pivoted repair of the ones that fail it (kernels_check.hip),
-m gpu.  The reference factors its patches with pivoted LU (LAPACK getrf through patch_pc_patch_dense_inverse,
alfi/solver.py:599-602; UMFPACK for Scott-Vogelius, :655-659); the fast inversion kernels do not pivot, so every
alfi_patches_factor checks || A_p X_p e - e || for ALL patches and re-inverts the flagged ones with partial pivoting."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.fixture(scope="module")
def ctx():
    from alfi_amd import hip
    c = hip.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("case", ["2d-P2", "3d-P1FB", "3d-P2FB"])
def test_probe_is_clean_on_the_shipped_problems(ctx, case):
    from alfi_amd import hip
    from alfi_amd.problem import TwoDimLidDrivenCavityProblem, ThreeDimLidDrivenCavityProblem, build_hierarchy
    mk, k, Re = {"2d-P2": (lambda: TwoDimLidDrivenCavityProblem(8), 2, 100.0),
                 "3d-P1FB": (lambda: ThreeDimLidDrivenCavityProblem(2), 1, 100.0),
                 "3d-P2FB": (lambda: ThreeDimLidDrivenCavityProblem(2), 2, 1000.0)}[case]
    lv, _ = build_hierarchy(mk(), 1, k, Re=Re)
    L = lv[-1]
    dl = hip.Level(ctx, L.A, L.bc_dofs)
    dl.set_patches(L.patch_ptr, L.patch_dofs)
    dl.factor()
    worst, flagged, repaired, after = dl.patch_check()
    # cond(A_p) ~ gamma / nu up to 1e7: residual of a backward-stable inverse ~ cond * eps * |e|
    assert 0.0 <= worst < 1e-7 and flagged == 0 and repaired == 0 and after == worst
    dl.close()


def _indefinite_level(seed=0):
    """A 2-D level operator with the sparsity of the real one but values no unpivoted elimination survives: every diagonal
    2 x 2 block is [[0, 4], [4, 0]] (+ small noise elsewhere), so the very first pivot of every patch is exactly zero,
    while the matrix itself is well conditioned."""
    from alfi_amd.problem import TwoDimLidDrivenCavityProblem, build_hierarchy, BSR
    lv, _ = build_hierarchy(TwoDimLidDrivenCavityProblem(4), 1, 2, Re=10.0)
    L = lv[-1]
    rng = np.random.default_rng(seed)
    A = L.A
    vals = 0.05 * rng.standard_normal(A.vals.shape)
    rows = np.repeat(np.arange(A.nbrows), np.diff(A.rowptr))
    diag = rows == A.colidx
    vals[diag] = np.array([[0.0, 4.0], [4.0, 0.0]]) + 0.05 * rng.standard_normal((int(diag.sum()), 2, 2))
    vals[diag, 0, 0] = 0.0
    return L, BSR(A.nbrows, A.nbcols, 2, A.rowptr, A.colidx, vals)


@pytest.mark.parametrize("tiny", [0.0, 1e-13])
def test_flagged_patches_are_reinverted_with_pivoting(ctx, tiny):
    from alfi_amd import hip
    L, A = _indefinite_level()
    if tiny:
        rows = np.repeat(np.arange(A.nbrows), np.diff(A.rowptr))
        A.vals[rows == A.colidx, 0, 0] = tiny                  # not a zero pivot: a tiny one, i.e. unbounded growth
    S = A.to_scipy().tocsr()
    dl = hip.Level(ctx, A, np.zeros(0, dtype=np.int32))
    dl.set_patches(L.patch_ptr, L.patch_dofs)
    dl.factor()                                               # must not raise: the pivoted path takes over
    worst, flagged, repaired, after = dl.patch_check()
    npatch = len(L.patch_ptr) - 1
    assert flagged == npatch == repaired and not (worst < 1e-6) and after < 1e-10
    n = np.diff(L.patch_ptr)
    for p in range(0, npatch, 5):
        dofs = L.patch_dofs[L.patch_ptr[p]:L.patch_ptr[p + 1]]
        ref = np.linalg.inv(S[dofs][:, dofs].toarray())
        assert relerr(dl.patch_inverse(p, int(n[p])), ref) < 1e-11
    x = np.random.default_rng(1).standard_normal(L.n)
    dx, dy = ctx.vec(x), ctx.vec(L.n)
    dl.patch_apply(dx, dy)
    y = np.zeros(L.n)
    for p in range(npatch):
        dofs = L.patch_dofs[L.patch_ptr[p]:L.patch_ptr[p + 1]]
        y[dofs] += np.linalg.solve(S[dofs][:, dofs].toarray(), x[dofs])
    assert relerr(dy.get(), y) < 1e-11
    dl.close()


def test_singular_patch_is_reported(ctx):
    from alfi_amd import hip
    L, A = _indefinite_level()
    dofs = L.patch_dofs[L.patch_ptr[3]:L.patch_ptr[4]]
    node = dofs[0] // 2
    A.vals[A.rowptr[node]:A.rowptr[node + 1]] = 0.0           # a zero row inside patch 3: singular whatever the pivoting
    dl = hip.Level(ctx, A, np.zeros(0, dtype=np.int32))
    dl.set_patches(L.patch_ptr, L.patch_dofs)
    with pytest.raises(hip.AlfiHipError, match="singular"):
        dl.factor()
    dl.close()


def test_newton_jacobian_with_supg_at_re_5000(ctx):
    """The Jacobian the authors' production runs factor (examples/generate_submission:18-20: --stabilisation-type supg,
    continuation to Re 5000): ldc3d [P2+FB]^3, linearised about the lid-driven state with the SUPG term, every patch
    probed; patches the unpivoted kernel cannot handle must have been repaired."""
    from alfi_amd import hip, _hostlib
    from alfi_amd.problem import ThreeDimLidDrivenCavityProblem, build_hierarchy, BSR
    prob = ThreeDimLidDrivenCavityProblem(2)
    lv, _ = build_hierarchy(prob, 2, 2, Re=5000.0)
    L = lv[-1]
    V = L.V
    state = 4.0 * np.ascontiguousarray(prob.driver(V.node_coords))      # a strong wind: cell Reynolds number >> 1
    g, vol = V.mesh.cell_geometry()
    vals = _hostlib.assemble_bsr(V.cell_nodes, g, vol, V.element.reference_tensors(), V.dim, L.A.rowptr, L.A.colidx,
                                 nu=L.nu, gamma=L.gamma, adv=1.0, wind=state)
    _hostlib.supg(V, state, L.nu, 0.05, 9.0, L.A.rowptr, L.A.colidx, vals)
    _hostlib.apply_bc_bsr(V.num_nodes, V.dim, L.A.rowptr, L.A.colidx, vals, np.repeat(V.bc_node_mask, V.dim))
    A = BSR(L.A.nbrows, L.A.nbcols, L.bs, L.A.rowptr, L.A.colidx, vals)
    dl = hip.Level(ctx, A, L.bc_dofs)
    dl.set_patches(L.patch_ptr, L.patch_dofs)
    dl.factor()
    worst, flagged, repaired, after = dl.patch_check()
    assert flagged == repaired and after < 1e-6
    S = A.to_scipy().tocsr()
    n = np.diff(L.patch_ptr)
    for p in np.random.default_rng(0).choice(len(n), 8, replace=False):
        dofs = L.patch_dofs[L.patch_ptr[p]:L.patch_ptr[p + 1]]
        Ap = S[dofs][:, dofs].toarray()
        X = dl.patch_inverse(p, int(n[p]))
        assert np.abs(Ap @ X - np.eye(len(dofs))).max() < 1e-7
    print("SUPG Re 5000: worst probe residual %.3e, %d of %d patches repaired" % (worst, repaired, len(n)))
    dl.close()
