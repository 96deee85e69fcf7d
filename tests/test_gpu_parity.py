"""Parity of the HIP path (through the C ABI) against the oracle, on the same seeded inputs.  -m gpu.

Floating-point tolerances (FP64 everywhere).  The patch operators have condition numbers up to ~1e7 (gamma/nu), so two
backward-stable inversions (LAPACK getrf/getri in the oracle, unpivoted register Gauss-Jordan on the GPU) agree to about
cond * eps ~ 1e-9 relative in anything that applies the inverses; summation order differs as well.  Tolerances below:
SpMV 1e-13, patch apply / smoother / transfers 1e-7 relative to the max-norm of the oracle result.  Whole cycles chain
2k..2k(L+1) smoother calls whose FGMRES least-squares problems amplify those 1e-10-level differences: perturbing the
ORACLE's own patch inverses by 1e-10 relative changes its V- and F-cycle output by 4e-6 (measured, 3-D P2+FB, Re 1000),
so cycles are compared at 1e-5 and, in addition, through their residual norms."""
CYCLE_TOL = 1e-5
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from alfi_amd.problem import (TwoDimLidDrivenCavityProblem, ThreeDimLidDrivenCavityProblem, build_hierarchy)

CASES = [("2d-P2", lambda: TwoDimLidDrivenCavityProblem(4), 2, 2, 100.0, 6),
         ("3d-P1FB", lambda: ThreeDimLidDrivenCavityProblem(2), 1, 1, 100.0, 4),
         ("3d-P2FB", lambda: ThreeDimLidDrivenCavityProblem(2), 2, 1, 1000.0, 4)]


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.fixture(scope="module")
def ctx():
    from alfi_amd import hip
    c = hip.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module", params=CASES, ids=[c[0] for c in CASES])
def setup(request, ctx):
    from alfi_amd import hip
    from oracle import alfi_oracle as O
    name, mk, k, nref, Re, ksm = request.param
    lv, tr = build_hierarchy(mk(), nref, k, Re=Re)
    omg = {sr: O.build_oracle_mg(lv, tr, ksm, schoeberl_restriction=sr) for sr in (False, True)}
    dmg = {sr: hip.Multigrid(ctx, lv, tr, ksm, robust_restriction=sr) for sr in (False, True)}
    yield dict(lv=lv, tr=tr, omg=omg, dmg=dmg, k=ksm, name=name)
    for m in dmg.values():
        m.close()


def rhs(n, bc, seed=0):
    b = np.random.default_rng(seed).standard_normal(n)
    b[bc] = 0.0
    return b


def test_spmv_and_residual(ctx, setup):
    for L, dl, ol in zip(setup["lv"], setup["dmg"][False].levels, setup["omg"][False].levels):
        x, b = rhs(L.n, [], 1), rhs(L.n, [], 2)
        dx, db, dy = ctx.vec(x), ctx.vec(b), ctx.vec(L.n)
        dl.spmv(dx, dy)
        assert relerr(dy.get(), ol["A"] @ x) < 1e-13
        dl.residual(db, dx, dy)
        assert relerr(dy.get(), b - ol["A"] @ x) < 1e-13


def test_patch_inverses(ctx, setup):
    L, dl, ol = setup["lv"][-1], setup["dmg"][False].levels[-1], setup["omg"][False].levels[-1]
    n = np.diff(L.patch_ptr)
    worst = 0.0
    for p in range(0, len(n), max(1, len(n) // 40)):
        worst = max(worst, relerr(dl.patch_inverse(p, int(n[p])), ol["smoother"].inv[p]))
    assert worst < 1e-7
    npatch, sum_n, sum_n2 = dl.patch_stats()
    assert (npatch, sum_n, sum_n2) == (len(n), n.sum(), (n * n).sum())


def test_patch_apply(ctx, setup):
    for L, dl, ol in list(zip(setup["lv"], setup["dmg"][False].levels, setup["omg"][False].levels))[1:]:
        x = rhs(L.n, [], 3)          # Dirichlet entries non-zero on purpose: y[bc] must equal x[bc]
        dx, dy = ctx.vec(x), ctx.vec(L.n)
        dl.patch_apply(dx, dy)
        y = dy.get()
        ref = ol["smoother"].apply(x)
        assert relerr(y, ref) < 1e-7
        assert np.array_equal(y[L.bc_dofs], x[L.bc_dofs])
        assert np.array_equal(dx.get(), x)                    # x untouched
        dl.patch_apply(dx, dy)                                 # deterministic: bitwise reproducible
        assert np.array_equal(dy.get(), y)


def test_fgmres_smoother(ctx, setup):
    from oracle import alfi_oracle as O
    k = setup["k"]
    for L, dl, ol in list(zip(setup["lv"], setup["dmg"][False].levels, setup["omg"][False].levels))[1:]:
        b, x0 = rhs(L.n, L.bc_dofs, 4), rhs(L.n, L.bc_dofs, 5)
        for nonzero in (True, False):
            dx, db = ctx.vec(x0), ctx.vec(b)
            dl.smooth(k, db, dx, nonzero_guess=nonzero)
            ref = O.fgmres(lambda v: ol["A"] @ v, ol["smoother"].apply, b, x0 if nonzero else np.zeros_like(x0), k,
                           nonzero_guess=nonzero)
            assert relerr(dx.get(), ref) < 1e-7
    # zero right-hand side and zero guess: iterate stays zero (no NaN from 0/0)
    dx, db = ctx.vec(L.n), ctx.vec(L.n)
    dl.smooth(k, db, dx, nonzero_guess=True)
    assert np.array_equal(dx.get(), np.zeros(L.n))


def test_transfers(ctx, setup):
    lv = setup["lv"]
    for i, (dt, ot) in enumerate(zip(setup["dmg"][True].transfers, setup["omg"][True].transfers)):
        Lc, Lf = lv[i], lv[i + 1]
        xc, rf = rhs(Lc.n, Lc.bc_dofs, 6), rhs(Lf.n, [], 7)
        dxc, dxf, drf, drc = ctx.vec(xc), ctx.vec(Lf.n), ctx.vec(rf), ctx.vec(Lc.n)
        dt.prolong(dxc, dxf)
        ref = ot.st.prolong(xc)
        ref[Lf.bc_dofs] = 0
        assert relerr(dxf.get(), ref) < 1e-7
        dt.restrict(drf, drc, robust=True)
        ref = ot.st.restrict(rf)
        ref[Lc.bc_dofs] = 0
        assert relerr(drc.get(), ref) < 1e-7
        assert np.array_equal(drf.get(), rf)        # the reference's in-place division of its input is not reproduced
        dt.restrict(drf, drc, robust=False)
        ref = ot.PT_plain @ rf
        ref[Lc.bc_dofs] = 0
        assert relerr(drc.get(), ref) < 1e-12
        # adjointness on the device: <P~ u, r> == <u, R~ r> for u, r vanishing on the Dirichlet dofs
        rf0 = rhs(Lf.n, Lf.bc_dofs, 11)
        drf.set(rf0)
        dt.restrict(drf, drc, robust=True)
        lhs, rhs_ = dxf.get() @ rf0, xc @ drc.get()
        assert abs(lhs - rhs_) < 1e-9 * max(abs(lhs), 1.0)


@pytest.mark.parametrize("robust", [False, True])
def test_vcycle_and_fcycle(ctx, setup, robust):
    lv = setup["lv"]
    L = lv[-1]
    omg, dmg = setup["omg"][robust], setup["dmg"][robust]
    b = rhs(L.n, L.bc_dofs, 8)
    db, dx = ctx.vec(b), ctx.vec(L.n)
    dmg.vcycle(db, dx)
    ref = omg.vcycle(len(lv) - 1, b, np.zeros(L.n))
    assert relerr(dx.get(), ref) < CYCLE_TOL
    # second cycle from the first iterate: residual norms match
    dmg.vcycle(db, dx)
    ref2 = omg.vcycle(len(lv) - 1, b, ref)
    A = omg.levels[-1]["A"]
    r_dev, r_ref = np.linalg.norm(b - A @ dx.get()), np.linalg.norm(b - A @ ref2)
    assert abs(r_dev - r_ref) < 1e-6 * np.linalg.norm(b)
    assert r_ref < 0.5 * np.linalg.norm(b)            # and the cycle actually converges
    dxf = ctx.vec(L.n)
    dmg.fcycle(db, dxf)
    ref_f = omg.fcycle(b)
    assert relerr(dxf.get(), ref_f) < CYCLE_TOL
    rf_dev, rf_ref = np.linalg.norm(b - A @ dxf.get()), np.linalg.norm(b - A @ ref_f)
    assert abs(rf_dev - rf_ref) < 1e-6 * np.linalg.norm(b)


def test_operator_update_refactors(ctx, setup):
    """PatchPC.update: new values, same sparsity -> patches must be refactored before the next apply."""
    from alfi_amd import hip
    L = setup["lv"][-1]
    dl = hip.Level(ctx, L.A, L.bc_dofs)
    dl.set_patches(L.patch_ptr, L.patch_dofs)
    x, y = ctx.vec(rhs(L.n, [], 9)), ctx.vec(L.n)
    with pytest.raises(hip.AlfiHipError):
        dl.patch_apply(x, y)                      # not factored yet
    dl.factor()
    dl.patch_apply(x, y)
    y1 = y.get()
    dl.update_values(2.0 * L.A.vals)
    with pytest.raises(hip.AlfiHipError):
        dl.patch_apply(x, y)
    dl.factor()
    dl.patch_apply(x, y)
    y2 = y.get()
    free = np.setdiff1d(np.arange(L.n), L.bc_dofs)
    assert relerr(y2[free], 0.5 * y1[free]) < 1e-9
    dl.close()


def test_bad_arguments_are_reported(ctx, setup):
    from alfi_amd import hip
    L = setup["lv"][-1]
    dl = hip.Level(ctx, L.A, L.bc_dofs)
    with pytest.raises(hip.AlfiHipError):
        dl.set_patches(np.array([0, 2]), np.array([5, 5]))            # not strictly ascending
    with pytest.raises(hip.AlfiHipError):
        dl.set_patches(np.array([0, 1]), np.array([L.n]))             # out of range
    with pytest.raises(hip.AlfiHipError):
        dl.set_patches(np.array([0, 4097]), np.arange(4097) % L.n)    # too large (limit: 4096 dofs per patch)
    dl.close()


@pytest.mark.parametrize("bs", [2, 3])
def test_spmv_segmented_kernel_on_ragged_rows(ctx, bs):
    """The nnz-balanced SpMV (csrc/kernels_vec.hip: bsr_spmv_flat_kernel) on block rows of very different lengths: rows
    shorter than a wave, rows spanning several 64-block iterations and rows spanning several 256-block chunks (carry +
    fix-up path), for y = A x and r = b - A x."""
    import scipy.sparse as sp
    from alfi_amd import hip
    from alfi_amd.problem import BSR
    rng = np.random.default_rng(5)
    nb = 1500
    rows, cols = [], []
    for i in range(nb):
        cnt = 900 if i % 97 == 0 else (260 if i % 31 == 0 else int(rng.integers(1, 70)))
        c = np.unique(np.concatenate([[i], rng.integers(0, nb, cnt)]))
        rows.append(np.full(c.shape[0], i))
        cols.append(c)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    blocks = rng.standard_normal((rows.shape[0], bs, bs))
    order = np.lexsort((cols, rows))
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=nb))])
    A = BSR(nb, nb, bs, rowptr, cols[order], blocks[order])
    S = A.to_scipy().tocsr()
    lvl = hip.Level(ctx, A, np.zeros(0, dtype=np.int32))
    x, b = rng.standard_normal(nb * bs), rng.standard_normal(nb * bs)
    dx, db, dy = ctx.vec(x), ctx.vec(b), ctx.vec(nb * bs)
    lvl.spmv(dx, dy)
    ref = S @ x
    assert np.abs(dy.get() - ref).max() / np.abs(ref).max() < 1e-13
    lvl.residual(db, dx, dy)
    assert np.abs(dy.get() - (b - ref)).max() / np.abs(ref).max() < 1e-13
    lvl.close()


@pytest.mark.parametrize("bs", [2, 3])
def test_patch_sizes_1_to_160_all_row_piece_combinations(ctx, bs):
    """Ragged synthetic patches of every size the library accepts (n_p = bs .. 160, i.e. every combination of the row
    pieces 128/64/32/16/8/4/2 of the inverse layout, odd n_p with the zero padding row, single-node patches, nodes shared
    by many patches and nodes in no patch) on a random diagonally dominant block operator: inverses, additive apply and a
    multiplicative sweep against NumPy; the empty patch set; sizes above 160 are refused."""
    import scipy.sparse as sp
    from alfi_amd import hip
    from alfi_amd.problem import BSR
    rng = np.random.default_rng(bs)
    nb = 400
    M = sp.random(nb * bs, nb * bs, density=0.02, random_state=7, format="csr")
    M = M + M.T + sp.identity(nb * bs) * 25.0
    A = BSR.from_scipy(sp.csr_matrix(M), bs)
    S = A.to_scipy().tocsr()
    max_nodes = 160 // bs
    sizes = list(range(1, max_nodes + 1)) + [1, 2, max_nodes, max_nodes]
    ptr, dofs = [0], []
    for sz in sizes:
        nodes = np.sort(rng.choice(nb - 5, sz, replace=False))        # the last 5 nodes are in no patch
        d = (nodes[:, None] * bs + np.arange(bs)).ravel()
        dofs.append(d)
        ptr.append(ptr[-1] + len(d))
    ptr, dofs = np.array(ptr, dtype=np.int64), np.concatenate(dofs).astype(np.int32)
    bc = np.array([0, 1], dtype=np.int32)
    lvl = hip.Level(ctx, A, bc)
    lvl.set_patches(ptr, dofs)
    lvl.factor()
    x = rng.standard_normal(nb * bs)
    ref = np.zeros_like(x)
    invs = []
    for p in range(len(sizes)):
        d = dofs[ptr[p]:ptr[p + 1]]
        Ainv = np.linalg.inv(S[d][:, d].toarray())
        invs.append(Ainv)
        got = lvl.patch_inverse(p, len(d))
        assert np.abs(got - Ainv).max() < 1e-11 * np.abs(Ainv).max(), (p, len(d))
        ref[d] += Ainv @ x[d]
    ref[bc] = x[bc]
    dx, dy = ctx.vec(x), ctx.vec(nb * bs)
    lvl.patch_apply(dx, dy)
    assert np.abs(dy.get() - ref).max() < 1e-12 * np.abs(ref).max()
    # multiplicative sweep in a scrambled order, symmetrised: first with every patch (patches of more than 64 nodes switch the
    # level to the workgroup-per-patch sweep kernel), then with the patches of at most 64 nodes (a wave per patch)
    Sc = S.tocsc()

    def sweep_check():
        order = rng.permutation(len(sizes))
        nw = lvl.set_multiplicative(order, True)
        assert 1 <= nw <= len(sizes)
        y, r = np.zeros_like(x), x.copy()
        for p in list(order) + list(order[::-1]):
            d = dofs[ptr[p]:ptr[p + 1]]
            dyp = invs[p] @ r[d]
            y[d] += dyp
            r -= Sc[:, d] @ dyp
        y[bc] = x[bc]
        lvl.patch_apply(dx, dy)
        assert np.abs(dy.get() - y).max() < 1e-11 * np.abs(y).max()
    sweep_check()
    if max(sizes) > 64:
        keep = [p for p, sz in enumerate(sizes) if sz <= 64]
        dofs = np.concatenate([dofs[ptr[p]:ptr[p + 1]] for p in keep]).astype(np.int32)
        invs = [invs[p] for p in keep]
        sizes = [sizes[p] for p in keep]
        ptr = np.concatenate([[0], np.cumsum([s_ * bs for s_ in sizes])]).astype(np.int64)
        lvl.set_patches(ptr, dofs)
        lvl.factor()
        sweep_check()
    lvl.set_multiplicative(None, False)
    # empty patch set: the smoother reduces to the Dirichlet copy
    lvl.set_patches(np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int32))
    lvl.factor()
    lvl.patch_apply(dx, dy)
    e = np.zeros_like(x)
    e[bc] = x[bc]
    assert np.array_equal(dy.get(), e)
    # one patch too large (the limit is 4096 dofs, tests/test_gpu_parity.py::test_large_patches covers 161 .. 2048)
    big = (np.arange(4096 // bs + 1)[:, None] * bs + np.arange(bs)).ravel().astype(np.int32) % (nb * bs)
    with pytest.raises(hip.AlfiHipError):
        lvl.set_patches(np.array([0, len(big)], dtype=np.int64), big)
    lvl.close()


@pytest.mark.parametrize("bs,max_n", [(2, 16), (2, 32), (3, 15), (3, 30)])
def test_small_patch_kernel_ragged_sizes(ctx, bs, max_n):
    """Levels whose patches all have n_p <= 32 take the lane-group kernel (patch_apply_small_kernel: 8 or 16 lanes per
    patch): every size bs .. max_n, more patches than fit a wave, against NumPy."""
    import scipy.sparse as sp
    from alfi_amd import hip
    from alfi_amd.problem import BSR
    rng = np.random.default_rng(100 + max_n)
    nb = 300
    M = sp.random(nb * bs, nb * bs, density=0.03, random_state=3, format="csr")
    M = M + M.T + sp.identity(nb * bs) * 20.0
    A = BSR.from_scipy(sp.csr_matrix(M), bs)
    S = A.to_scipy().tocsr()
    sizes = [int(s_) for s_ in rng.integers(1, max_n // bs + 1, 700)] + list(range(1, max_n // bs + 1))
    ptr, dofs = [0], []
    for sz in sizes:
        nodes = np.sort(rng.choice(nb, sz, replace=False))
        d = (nodes[:, None] * bs + np.arange(bs)).ravel()
        dofs.append(d)
        ptr.append(ptr[-1] + len(d))
    ptr, dofs = np.array(ptr, dtype=np.int64), np.concatenate(dofs).astype(np.int32)
    lvl = hip.Level(ctx, A, np.zeros(0, dtype=np.int32))
    lvl.set_patches(ptr, dofs)
    lvl.factor()
    x = rng.standard_normal(nb * bs)
    ref = np.zeros_like(x)
    for p in range(len(sizes)):
        d = dofs[ptr[p]:ptr[p + 1]]
        ref[d] += np.linalg.solve(S[d][:, d].toarray(), x[d])
    dx, dy = ctx.vec(x), ctx.vec(nb * bs)
    lvl.patch_apply(dx, dy)
    assert np.abs(dy.get() - ref).max() < 1e-12 * np.abs(ref).max()
    lvl.close()


def test_uncommon_paths(ctx):
    """k = 20 smoothing iterations (Gram-Schmidt passes of more than 16 vectors), k = 1, a one-level hierarchy (coarse
    solve only), repeated operator updates with refactorisation."""
    from alfi_amd import hip
    from oracle import alfi_oracle as O
    lv, tr = build_hierarchy(TwoDimLidDrivenCavityProblem(4), 1, 2, Re=10.0)
    L = lv[-1]
    b = rhs(L.n, L.bc_dofs, 3)
    for k in (1, 20):
        omg = O.build_oracle_mg(lv, tr, k)
        dmg = hip.Multigrid(ctx, lv, tr, k)
        db, dx = ctx.vec(b), ctx.vec(L.n)
        dmg.vcycle(db, dx)
        ref = omg.vcycle(len(lv) - 1, b, np.zeros(L.n))
        assert relerr(dx.get(), ref) < 1e-7, k
        x0 = rhs(L.n, L.bc_dofs, 4)
        dx0 = ctx.vec(x0)
        dmg.levels[-1].smooth(k, db, dx0, nonzero_guess=True)
        assert relerr(dx0.get(), omg.smooth(len(lv) - 1, b, x0)) < 1e-7, k
        dmg.close()
    # one level: the cycle is the coarse solve
    one = hip.Multigrid(ctx, lv[:1], [], 3)
    b0 = rhs(lv[0].n, lv[0].bc_dofs, 5)
    d0, x0 = ctx.vec(b0), ctx.vec(lv[0].n)
    one.vcycle(d0, x0)
    sol = np.linalg.solve(lv[0].A.to_scipy().toarray(), b0)
    assert relerr(x0.get(), sol) < 1e-10
    one.fcycle(d0, x0)
    assert relerr(x0.get(), sol) < 1e-10
    one.close()
    # update -> factor -> apply, twice, tracks the new operator each time
    dl = hip.Level(ctx, L.A, L.bc_dofs)
    dl.set_patches(L.patch_ptr, L.patch_dofs)
    x = rhs(L.n, [], 6)
    dxx, dyy = ctx.vec(x), ctx.vec(L.n)
    for scale in (1.0, 3.0, 0.5):
        dl.update_values(scale * L.A.vals)
        dl.factor()
        dl.patch_apply(dxx, dyy)
        A = (scale * L.A.to_scipy()).tocsr()
        ref = O.PatchSmoother(A, L.patch_ptr, L.patch_dofs, L.bc_dofs).apply(x)
        assert relerr(dyy.get(), ref) < 1e-9
        dl.spmv(dxx, dyy)
        assert relerr(dyy.get(), A @ x) < 1e-13
    dl.close()


@pytest.mark.parametrize("bs", [2, 3])
def test_large_patches(ctx, bs):
    """Patches beyond the register-resident path (161 .. ~1300 dofs; the macro-star sizes of SURVEY.md section 8: 405,
    1275): gather, blocked Gauss-Jordan inversion with the FP64 matrix cores, one-workgroup-per-patch apply -- against
    NumPy on a random diagonally dominant block operator.  Sizes straddle the 64-column blocks and the 128-row pieces."""
    import scipy.sparse as sp
    from alfi_amd import hip
    from alfi_amd.problem import BSR
    rng = np.random.default_rng(50 + bs)
    nb = 900
    M = sp.random(nb * bs, nb * bs, density=0.01, random_state=11, format="csr")
    M = M + M.T + sp.identity(nb * bs) * 30.0
    A = BSR.from_scipy(sp.csr_matrix(M), bs)
    S = A.to_scipy().tocsr()
    sizes = [(161 + bs - 1) // bs, 200 // bs, 64, 405 // bs, 640 // bs, 1275 // bs, 1300 // bs, 10]   # nodes per patch
    ptr, dofs = [0], []
    for sz in sizes:
        nodes = np.sort(rng.choice(nb, sz, replace=False))
        d = (nodes[:, None] * bs + np.arange(bs)).ravel()
        dofs.append(d)
        ptr.append(ptr[-1] + len(d))
    ptr, dofs = np.array(ptr, dtype=np.int64), np.concatenate(dofs).astype(np.int32)
    lvl = hip.Level(ctx, A, np.array([0], dtype=np.int32))
    lvl.set_patches(ptr, dofs)
    lvl.factor()
    x = rng.standard_normal(nb * bs)
    ref = np.zeros_like(x)
    for p in range(len(sizes)):
        d = dofs[ptr[p]:ptr[p + 1]]
        Ainv = np.linalg.inv(S[d][:, d].toarray())
        got = lvl.patch_inverse(p, len(d))
        assert np.abs(got - Ainv).max() < 1e-10 * np.abs(Ainv).max(), (p, len(d))
        ref[d] += Ainv @ x[d]
    ref[0] = x[0]
    dx, dy = ctx.vec(x), ctx.vec(nb * bs)
    lvl.patch_apply(dx, dy)
    assert np.abs(dy.get() - ref).max() < 1e-11 * np.abs(ref).max()
    # as smoother of FGMRES
    b = rng.standard_normal(nb * bs)
    b[0] = 0.0
    db, dz = ctx.vec(b), ctx.vec(nb * bs)
    lvl.smooth(3, db, dz, nonzero_guess=False)
    from oracle import alfi_oracle as O
    sm = O.PatchSmoother(S, ptr, dofs, np.array([0]))
    zref = O.fgmres(lambda v: S @ v, sm.apply, b, np.zeros_like(b), 3, nonzero_guess=False)
    assert np.abs(dz.get() - zref).max() < 1e-8 * np.abs(zref).max()
    lvl.close()


@pytest.mark.parametrize("composition", ["additive", "multiplicative"])
def test_cycles_replayed_as_hip_graphs(ctx, composition):
    """alfi_ctx_set_graph: the captured V- and full cycles give bit-identical results to the eager ones, pick up new
    operator values written in place, and re-capture when (nu, gamma) of a transfer change."""
    from alfi_amd import hip
    from alfi_amd.problem import TwoDimLidDrivenCavityProblem, build_hierarchy
    from alfi_amd.relaxation import OrderedRelaxation, Options
    lv, tr = build_hierarchy(TwoDimLidDrivenCavityProblem(4), 2, 2, Re=100.0)
    mg = hip.Multigrid(ctx, lv, tr, 3, robust_restriction=True)
    if composition == "multiplicative":
        for Lh, dl in zip(lv[1:], mg.levels[1:]):
            orl = OrderedRelaxation()
            orl.name = "Star"
            orl.opts = Options("", {"pc_patch_construction_Star_sort_order": "0+:1-"})
            dl.set_multiplicative(orl.iteration_order(Lh.V.mesh.coords[Lh.patch_seeds]), True)
    L = lv[-1]
    rng = np.random.default_rng(5)
    b = rng.standard_normal(L.n)
    b[L.bc_dofs] = 0
    db, dx = ctx.vec(b), ctx.vec(L.n)

    def run(cycle, reps):
        outs = []
        for _ in range(reps):
            dx.zero()
            cycle(db, dx)
            outs.append(dx.get())
        return outs
    ev, ef = run(mg.vcycle, 1)[0], run(mg.fcycle, 1)[0]
    ctx.set_graph(True)
    try:
        gv = run(mg.vcycle, 4)          # eager, capture + replay, replay, replay
        gf = run(mg.fcycle, 3)
        assert all(np.array_equal(g, ev) for g in gv) and all(np.array_equal(g, ef) for g in gf)
        # new operator values in place: the replay must see them
        vals = L.A.vals * 1.5
        mg.levels[-1].update_values(vals)
        mg.levels[-1].factor()
        g2 = run(mg.vcycle, 2)
        ctx.set_graph(False)
        e2 = run(mg.vcycle, 1)[0]
        assert np.array_equal(g2[0], e2) and np.array_equal(g2[1], e2) and not np.array_equal(e2, ev)
        # new transfer parameters: signature changes, the cycle is captured again
        ctx.set_graph(True)
        mg.transfers[-1].update(L.nu * 2, L.gamma)
        g3 = run(mg.vcycle, 3)
        ctx.set_graph(False)
        e3 = run(mg.vcycle, 1)[0]
        assert all(np.array_equal(g, e3) for g in g3) and not np.array_equal(e3, e2)
    finally:
        ctx.set_graph(False)
        mg.close()


def test_cycle_logic_with_the_device_inverses(ctx, setup):
    """Separates inversion rounding from cycle logic: the oracle gets the DEVICE's patch inverses (alfi_patch_get_inverse)
    and interior-block inverses (alfi_transfer_get_block_inverse) in place of its LAPACK ones; what is left to differ is
    the order of floating-point sums.  Smoother then agrees to 1e-9, transfers, V- and F-cycles to 5e-9 -- against CYCLE_TOL = 1e-5
    with independently inverted patches -- so a wrong Givens rotation, Hessenberg column, restriction or coarse correction
    cannot hide inside the inversion tolerance."""
    from oracle import alfi_oracle as O
    lv, k = setup["lv"], setup["k"]
    dmg = setup["dmg"][True]
    omg = O.build_oracle_mg(lv, setup["tr"], k, schoeberl_restriction=True)     # a private copy to modify
    for L, dl, ol in list(zip(lv, dmg.levels, omg.levels))[1:]:
        n = np.diff(L.patch_ptr)
        ol["smoother"].inv = [dl.patch_inverse(p, int(n[p])) for p in range(len(n))]
    for dt, ot in zip(dmg.transfers, omg.transfers):
        m = ot.st.blk_dofs.shape[1]
        binv = [dt.block_inverse(b, m) for b in range(ot.st.blk_dofs.shape[0])]
        # the oracle solves with LU factors; feed it exact "LU factors" of the device inverse's action instead
        ot.st.lu = None
        ot.st._binv = binv

        def patch_apply(x, st=ot.st):
            y = np.zeros_like(x)
            for d, X in zip(st.blk_dofs, st._binv):
                y[d] = X @ x[d]
            y[st.skel] = x[st.skel]
            return y
        ot.st._patch_apply = patch_apply
    # the coarse solve: the explicit inverse the device multiplies with (hip.coarse_inverse), not the oracle's sparse LU
    from alfi_amd import hip
    Cinv = hip.coarse_inverse(lv[0].A)

    class _Coarse(object):
        @staticmethod
        def solve(v):
            return Cinv @ v
    omg.coarse_lu = _Coarse()
    tol = 1e-10
    L, dl, ol = lv[-1], dmg.levels[-1], omg.levels[-1]
    b, x0 = rhs(L.n, L.bc_dofs, 21), rhs(L.n, L.bc_dofs, 22)
    dx, db = ctx.vec(x0), ctx.vec(b)
    dl.smooth(k, db, dx, nonzero_guess=True)
    ref = O.fgmres(lambda v: ol["A"] @ v, ol["smoother"].apply, b, x0, k)
    es = relerr(dx.get(), ref)
    assert es < 1e-9            # measured 2e-11 .. 1.0e-10 (3d-P2FB, Re 1000: k dependent least-squares problems)
    for i, (dt, ot) in enumerate(zip(dmg.transfers, omg.transfers)):
        Lc, Lf = lv[i], lv[i + 1]
        xc, rf = rhs(Lc.n, Lc.bc_dofs, 23), rhs(Lf.n, [], 24)
        dxc, dxf, drf, drc = ctx.vec(xc), ctx.vec(Lf.n), ctx.vec(rf), ctx.vec(Lc.n)
        # interior blocks nu K + gamma D have condition numbers ~ gamma / nu (1e7 at Re 1000): the order of the 24 .. 27
        # terms of X b alone moves the result by cond * eps ~ 1e-9 (measured 8e-10 on 3d-P2FB)
        dt.prolong(dxc, dxf)
        r = ot.st.prolong(xc)
        r[Lf.bc_dofs] = 0
        assert relerr(dxf.get(), r) < 5e-9
        dt.restrict(drf, drc, robust=True)
        r = ot.st.restrict(rf)
        r[Lc.bc_dofs] = 0
        assert relerr(drc.get(), r) < 5e-9
    # whole cycles chain 2k .. 2k (L + 1) smoother iterations whose least-squares problems amplify the 1e-16 differences
    # of the summation orders: measured 8e-11 .. 9.3e-10 at Re 100 and 3e-9 .. 5.6e-8 at Re 1000 (3d-P2FB; a pure change
    # of summation order moves that case by 2e-8 .. 8e-8, scripts/ab_cycle.py) -- compared at 5e-9 / 5e-7, against
    # CYCLE_TOL = 1e-5 for independently inverted patches
    ctol = 5e-9 if setup["name"] != "3d-P2FB" else 5e-7      # Re 1000: measured up to 5.6e-8 (F-cycle)
    b = rhs(L.n, L.bc_dofs, 25)
    db, dx = ctx.vec(b), ctx.vec(L.n)
    dmg.vcycle(db, dx)
    ref = omg.vcycle(len(lv) - 1, b, np.zeros(L.n))
    e1 = relerr(dx.get(), ref)
    dmg.vcycle(db, dx)
    ref = omg.vcycle(len(lv) - 1, b, ref)
    e2 = relerr(dx.get(), ref)
    dmg.fcycle(db, dx)
    e3 = relerr(dx.get(), omg.fcycle(b))
    print("cycle logic with device inverses [%s]: smoother %.2e, V %.2e, second V %.2e, F %.2e"
          % (setup["name"], es, e1, e2, e3))
    assert max(e1, e2, e3) < ctol


def test_coarse_factorisation_by_the_library(ctx, setup):
    """alfi_coarse_factor (blocked Gauss-Jordan on the matrix cores + Newton-Schulz polish, no library GEMM, no host LAPACK)
    against a sparse direct solve (the reference: AssembledPC + LU, alfi/solver.py:369-378)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    from alfi_amd import hip
    L = setup["lv"][0]
    dl = hip.Level(ctx, L.A, L.bc_dofs)
    res = dl.coarse_factor()
    assert 0.0 <= res < 1e-8
    A = sp.csc_matrix(L.A.to_scipy())
    lu = spla.splu(A)
    b = rhs(L.n, L.bc_dofs, 31)
    db, dx = ctx.vec(b), ctx.vec(L.n)
    dl.coarse_solve(db, dx)
    err = relerr(dx.get(), lu.solve(b))
    print("coarse factorisation [%s]: n = %d, probe %.2e, vs splu %.2e" % (setup["name"], L.n, res, err))
    # two backward-stable solvers of a system with condition number ~1e7 agree to cond * eps ~ 1e-9 (measured 7e-10)
    assert err < 1e-8
    dl.close()
