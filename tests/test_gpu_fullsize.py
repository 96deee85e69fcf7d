"""BASELINE.json's configurations at FULL size on the GPU, checked through size-independent properties (the oracle needs
minutes to hours at these sizes):

* sampled patch inverses really invert the patch operators gathered from the host copy of the level operator;
* the additive patch smoother is linear and, on sampled dofs, equals the sum of its patch solves;
* the level SpMV agrees with SciPy on the full vector (configs 2, 3) / on sampled rows (config 4);
* Schoeberl prolongation and robust restriction are adjoint: <P~ xc, rf> = <xc, R~ rf> (transfer.py:186-192), and the
  prolongation keeps the discrete divergence of a divergence-free coarse field small (the property the transfer is built
  for, transfer.py:194-259);
* V-cycles contract the residual monotonically (PCMG, solver.py:359-368).

-m gpu.  Config 1 (N = 64) runs the same checks AND is compared with the oracle's V-cycle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CONFIGS = {
    # name: (dim, baseN, nref, element degree, Re, smoothing k)      BASELINE.json configs[0..3]
    "cfg1": (2, 16, 2, 2, 10.0, 6),
    "cfg2": (2, 6, 6, 2, 100.0, 6),
    "cfg3": (3, 2, 4, 1, 100.0, 10),
    "cfg4": (3, 7, 3, 2, 1000.0, 10),
}


def _build(name):
    from alfi_amd.problem import TwoDimLidDrivenCavityProblem, ThreeDimLidDrivenCavityProblem, build_hierarchy
    dim, baseN, nref, ke, Re, k = CONFIGS[name]
    prob = TwoDimLidDrivenCavityProblem(baseN) if dim == 2 else ThreeDimLidDrivenCavityProblem(baseN)
    lv, tr = build_hierarchy(prob, nref, ke, Re=Re)
    return lv, tr, k


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg3", "cfg4"])
def test_full_size_properties(name):
    from alfi_amd import hip
    lv, tr, k = _build(name)
    ctx = hip.Context(0)
    mg = hip.Multigrid(ctx, lv, tr, k, robust_restriction=True)
    assert 0.0 <= mg.levels[0].coarse_residual() < 1e-6        # residual probe of the library's own coarse inverse
    print("%s: coarse inverse probe %.2e (n = %d)" % (name, mg.levels[0].coarse_residual(), lv[0].n))
    L, dl = lv[-1], mg.levels[-1]
    n, bs = L.n, L.bs
    rng = np.random.default_rng(11)
    S = L.A.to_scipy().tocsr() if name != "cfg4" else None

    # -- patch inverses ------------------------------------------------------------------------------------------------------
    # ALL patches of every level passed the device-side residual probe || A_p X_p e - e || (alfi_patches_check)
    for dlev in mg.levels[1:]:
        worst, flagged, repaired, after = dlev.patch_check()
        assert 0.0 <= worst < 1e-6 and flagged == repaired, (name, worst, flagged, repaired)
    npatch = len(L.patch_ptr) - 1
    for p in rng.choice(npatch, 6, replace=False):
        dofs = L.patch_dofs[L.patch_ptr[p]:L.patch_ptr[p + 1]]
        if S is not None:
            Ap = S[dofs][:, dofs].toarray()
        else:
            Ap = L.A.select_rows(np.unique(dofs // bs)).to_scipy().tocsc()[:, dofs].toarray()
            rows = (np.searchsorted(np.unique(dofs // bs), dofs // bs) * bs + dofs % bs)
            Ap = Ap[rows]
        Ainv = dl.patch_inverse(p, len(dofs))
        res = np.abs(Ainv @ Ap - np.eye(len(dofs))).max()
        assert res < 1e-8 * np.linalg.cond(Ap), (p, res)

    # -- additive smoother: linear, and the sum of its patch solves on sampled dofs ------------------------------------------------
    x, z = rng.standard_normal(n), rng.standard_normal(n)
    dx, dz, dy = ctx.vec(x), ctx.vec(z), ctx.vec(n)
    dl.patch_apply(dx, dy)
    Mx = dy.get()
    dl.patch_apply(dz, dy)
    Mz = dy.get()
    dw = ctx.vec(2.0 * x - 3.0 * z)
    dl.patch_apply(dw, dy)
    lin = dy.get()
    assert np.abs(lin - (2.0 * Mx - 3.0 * Mz)).max() < 1e-9 * np.abs(lin).max()
    assert np.array_equal(Mx[L.bc_dofs], x[L.bc_dofs])
    # dof -> patches through a sample of patches: all patches containing the sampled patch's seed node
    seeds_nodes = L.V.vertex_nodes[L.patch_seeds]
    for p in rng.choice(npatch, 4, replace=False):
        d0 = int(seeds_nodes[p]) * bs            # the seed vertex' first dof lies in exactly one patch (its own star)
        dofs = L.patch_dofs[L.patch_ptr[p]:L.patch_ptr[p + 1]]
        row = np.flatnonzero(dofs == d0)
        assert row.size == 1
        val = dl.patch_inverse(p, len(dofs))[row[0]] @ x[dofs]
        assert abs(Mx[d0] - val) < 1e-9 * max(abs(val), np.abs(Mx).max() * 1e-3)

    # -- SpMV ----------------------------------------------------------------------------------------------------------------------------
    dl.spmv(dx, dy)
    Ax = dy.get()
    if S is not None:
        ref = S @ x
        assert np.abs(Ax - ref).max() < 1e-12 * np.abs(ref).max()
    else:
        nodes = rng.choice(L.A.nbrows, 2000, replace=False)
        ref = L.A.select_rows(nodes).to_scipy() @ x
        got = Ax.reshape(-1, bs)[nodes].ravel()
        assert np.abs(got - ref).max() < 1e-12 * np.abs(Ax).max()

    # -- transfers: adjointness; prolongation of a coarse field --------------------------------------------------------------------------------
    T, dt, Lc = tr[-1], mg.transfers[-1], lv[-2]
    xc = rng.standard_normal(Lc.n)
    xc[Lc.bc_dofs] = 0.0
    rf = rng.standard_normal(n)
    rf[L.bc_dofs] = 0.0
    dxc, dxf, drf, drc = ctx.vec(xc), ctx.vec(n), ctx.vec(rf), ctx.vec(Lc.n)
    dt.prolong(dxc, dxf)
    dt.restrict(drf, drc, robust=True)
    lhs, rhs = dxf.get() @ rf, xc @ drc.get()
    assert abs(lhs - rhs) < 1e-9 * max(abs(lhs), abs(rhs), 1.0)

    # -- V-cycles contract the residual ----------------------------------------------------------------------------------------------------------
    b = rng.standard_normal(n)
    b[L.bc_dofs] = 0.0
    db, du, dr = ctx.vec(b), ctx.vec(n), ctx.vec(n)
    hist = [np.linalg.norm(b)]
    for _ in range(3):
        mg.vcycle(db, du)
        dl.residual(db, du, dr)
        hist.append(np.linalg.norm(dr.get()))
    assert all(hist[i + 1] < 0.5 * hist[i] for i in range(3)), hist

    if name == "cfg1":
        # small enough for the oracle: the reference's own CPU-runnable configuration
        from oracle import alfi_oracle as O
        omg = O.build_oracle_mg(lv, tr, k, schoeberl_restriction=True)
        top = len(lv) - 1
        ref = np.zeros(n)
        for _ in range(3):
            ref = omg.vcycle(top, b, ref)
        got = du.get()
        assert np.abs(got - ref).max() < 1e-5 * np.abs(ref).max()
    mg.close()
    ctx.close()


def _bench_problem(name):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.build_problem(name, False)


@pytest.mark.parametrize("name", ["cfg5", "cfg6", "cfg5L", "cfg5m"])
def test_full_size_properties_of_the_widened_configs(name):
    """bench.py's configurations beyond BASELINE's first four at their bench sizes: config 5 on one GPU (bfs3d Scott-Vogelius
    [P3]^3, 441 k dofs, macro stars of up to 2175 dofs as CONDENSED factors), config 6 (ldc3d [P1+FB]^3 over the reference's
    largest coarse grid: 14.7 M dofs, multifrontal coarse solver) and config 5 with one more refinement (3.47 M dofs, 47 GB of
    condensed factors; 8 s of host generation since the macro-star constructor is vectorised, 281 s before), and config 5 on the
    reference's own gmsh channel (cfg5m: examples/bfs3d/coarse60.msh kept as an input fixture, 912 unstructured tets, one
    refinement: 440 022 dofs, macro stars of up to 3615 dofs, 56 784-dof coarse grid).  Size-independent properties:
    every patch passes the residual probe, the smoother is linear and copies Dirichlet entries, the level product agrees with
    SciPy on sampled rows, prolongation and robust restriction are adjoint, V-cycles contract the residual."""
    from alfi_amd import hip
    lv, tr, k = _bench_problem(name)
    ctx = hip.Context(0)
    mg = hip.Multigrid(ctx, lv, tr, k, robust_restriction=True)
    assert 0.0 <= mg.levels[0].coarse_residual() < 1e-6
    L, dl = lv[-1], mg.levels[-1]
    n, bs = L.n, L.bs
    rng = np.random.default_rng(17)
    for dlev in mg.levels[1:]:
        worst, flagged, repaired, after = dlev.patch_check()
        assert 0.0 <= worst < 1e-6 and flagged == repaired, (name, worst, flagged, repaired)
    if name.startswith("cfg5"):
        assert hip.condense_patches(L) and dl.factor_bytes() < 0.25 * 8 * dl.patch_stats()[2]    # condensed: < 1/4 of dense
    # smoother: linear, Dirichlet entries copied
    x, z = rng.standard_normal(n), rng.standard_normal(n)
    dx, dz, dy = ctx.vec(x), ctx.vec(z), ctx.vec(n)
    dl.patch_apply(dx, dy)
    Mx = dy.get()
    dl.patch_apply(dz, dy)
    Mz = dy.get()
    dl.patch_apply(ctx.vec(2.0 * x - 3.0 * z), dy)
    lin = dy.get()
    assert np.abs(lin - (2.0 * Mx - 3.0 * Mz)).max() < 1e-8 * np.abs(lin).max()
    assert np.array_equal(Mx[L.bc_dofs], x[L.bc_dofs])
    # the smoother inverts the operator on a patch: for y supported on ONE patch's dofs with A y = r (r on the patch only),
    # the patch's own contribution to M r is y -- checked through the residual probe above for every patch; here the product
    dl.spmv(dx, dy)
    Ax = dy.get()
    nodes = rng.choice(L.A.nbrows, 2000, replace=False)
    ref = L.A.select_rows(nodes).to_scipy() @ x
    assert np.abs(Ax.reshape(-1, bs)[nodes].ravel() - ref).max() < 1e-12 * np.abs(Ax).max()
    # transfers: adjointness
    dt, Lc = mg.transfers[-1], lv[-2]
    xc = rng.standard_normal(Lc.n)
    xc[Lc.bc_dofs] = 0.0
    rf = rng.standard_normal(n)
    rf[L.bc_dofs] = 0.0
    dxc, dxf, drf, drc = ctx.vec(xc), ctx.vec(n), ctx.vec(rf), ctx.vec(Lc.n)
    dt.prolong(dxc, dxf)
    dt.restrict(drf, drc, robust=True)
    lhs, rhs = dxf.get() @ rf, xc @ drc.get()
    assert abs(lhs - rhs) < 1e-8 * max(abs(lhs), abs(rhs), 1.0)
    # V-cycles contract the residual
    b = rng.standard_normal(n)
    b[L.bc_dofs] = 0.0
    db, du, dr = ctx.vec(b), ctx.vec(n), ctx.vec(n)
    hist = [np.linalg.norm(b)]
    for _ in range(3):
        mg.vcycle(db, du)
        dl.residual(db, du, dr)
        hist.append(np.linalg.norm(dr.get()))
    print(name, "residual history", hist)
    # (a random right-hand side on the Re 500 channel with its natural outflow contracts by ~0.3, 0.7, 0.85 per V-cycle --
    # the bench's 13 cycles reach 9e-3 -- where the cavity configs give < 0.5 every time: monotone + an overall bound)
    assert all(hist[i + 1] < 0.95 * hist[i] for i in range(3)) and hist[3] < 0.3 * hist[0], hist
    mg.close()
    ctx.close()
