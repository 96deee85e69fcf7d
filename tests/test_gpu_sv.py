"""Scott-Vogelius hierarchy on the GPU (-m gpu): macro-star patches (62 dofs) in the patch smoother, macro-cell blocks
(38 dofs > 32: solved with the patch kernels) in the Schoeberl transfer, full grad-div operators -- same library entry
points as PkP0.  Transfers, cycles and the gamma-robustness experiment against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from alfi_amd.problem import TwoDimLidDrivenCavityProblem
from alfi_amd.sv import build_sv_hierarchy
from tests.test_graddiv import fgmres_solve
from tests.test_sv import run as run_oracle


@pytest.mark.parametrize("case", ["2d-p2", "3d-p2", "3d-p3", "bfs3d-p3"])
def test_sv_transfers_and_cycles_match_oracle(case):
    """2-D: macro stars of 62 dofs, macro-cell blocks of 38.  3-D [P2]^3 (the smoother / transfer machinery, not an
    inf-sup stable pair): macro stars of up to 513 dofs (blocked matrix-core inversion), macro-cell blocks of 123 (odd).
    3-D [P3]^3 (the velocity space of BASELINE config 5): macro stars of up to 1599 dofs, macro-cell blocks of 390 --
    both through the blocked matrix-core inversion and the workgroup-per-patch apply.  bfs3d-p3: BASELINE config 5 at its
    smallest (bench.py --config cfg5s): backward-facing-step channel with a natural outflow, Re = 500, 57 k velocity dofs,
    303 macro stars of up to 1941 dofs."""
    from alfi_amd import hip
    from alfi_amd.problem import ThreeDimLidDrivenCavityProblem
    from oracle import alfi_oracle as O
    if case == "2d-p2":
        lv, tr = build_sv_hierarchy(TwoDimLidDrivenCavityProblem(2), 2, 2, Re=100.0, gamma=1e4)
    elif case == "3d-p2":
        lv, tr = build_sv_hierarchy(ThreeDimLidDrivenCavityProblem(1), 2, 2, Re=100.0, gamma=1e4)
        assert tr[0].blk_dofs.shape[1] == 123 and max(np.diff(L.patch_ptr).max() for L in lv[1:]) > 160
    elif case == "3d-p3":
        lv, tr = build_sv_hierarchy(ThreeDimLidDrivenCavityProblem(1), 1, 3, Re=100.0, gamma=1e4)
        assert tr[0].blk_dofs.shape[1] == 390 and np.diff(lv[1].patch_ptr).max() == 1599
    else:
        from alfi_amd.problem import ThreeDimBackwardsFacingStepProblem
        lv, tr = build_sv_hierarchy(ThreeDimBackwardsFacingStepProblem(1), 1, 3, Re=500.0, gamma=1e4)
        assert lv[1].n == 56937 and np.diff(lv[1].patch_ptr).max() == 1941
    ctx = hip.Context(0)
    k = 3
    if case == "bfs3d-p3":
        # the oracle needs ~1.5 min to invert these patches with LAPACK: its transfers and cycles of the seeded inputs are a
        # committed fixture (tests/golden/make_bfs3d.py; float32, compared at 1e-5 like every cycle)
        import os
        from tests.golden.make_bfs3d import inputs
        g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bfs3d_p3_cycles.npz"))
        uc, rf, b = inputs(lv)
        mg = hip.Multigrid(ctx, lv, tr, k, robust_restriction=True)
        duc, dxf, drf, drc = ctx.vec(uc), ctx.vec(lv[1].n), ctx.vec(rf), ctx.vec(lv[0].n)
        db, dx = ctx.vec(b), ctx.vec(lv[1].n)
        mg.transfers[0].prolong(duc, dxf)
        mg.transfers[0].restrict(drf, drc, robust=True)
        got = {"prolong": dxf.get(), "restrict": drc.get()}
        mg.vcycle(db, dx)
        got["vcycle"] = dx.get()
        mg.fcycle(db, dx)
        got["fcycle"] = dx.get()
        for key, v in got.items():
            ref = g[key].astype(np.float64)
            assert np.abs(v - ref).max() < 1e-5 * np.abs(ref).max(), key
            assert abs(np.linalg.norm(v) - float(g[key + "_norm"])) < 1e-5 * float(g[key + "_norm"]), key
        mg.close()
        ctx.close()
        return
    rng = np.random.default_rng(0)
    for robust in (True, False):
        mg = hip.Multigrid(ctx, lv, tr, k, robust_restriction=robust)
        omg = O.build_oracle_mg(lv, tr, k, schoeberl_restriction=robust)
        for l in range(1, len(lv)):
            uc = rng.standard_normal(lv[l - 1].n)
            uc[lv[l - 1].bc_dofs] = 0
            rf = rng.standard_normal(lv[l].n)
            duc, dxf, drf, drc = ctx.vec(uc), ctx.vec(lv[l].n), ctx.vec(rf), ctx.vec(lv[l - 1].n)
            mg.transfers[l - 1].prolong(duc, dxf)
            ref = omg.prolong(l, uc)
            assert np.abs(dxf.get() - ref).max() < 1e-7 * np.abs(ref).max()
            mg.transfers[l - 1].restrict(drf, drc, robust=robust)
            ref = omg.restrict(l, rf)
            assert np.abs(drc.get() - ref).max() < 1e-7 * np.abs(ref).max()
        L = lv[-1]
        b = rng.standard_normal(L.n)
        b[L.bc_dofs] = 0
        db, dx = ctx.vec(b), ctx.vec(L.n)
        mg.vcycle(db, dx)
        ref = omg.vcycle(len(lv) - 1, b, np.zeros(L.n))
        assert np.abs(dx.get() - ref).max() < 1e-5 * np.abs(ref).max()
        mg.fcycle(db, dx)
        ref = omg.fcycle(b)
        assert np.abs(dx.get() - ref).max() < 1e-5 * np.abs(ref).max()
        mg.close()
    ctx.close()


def test_sv_gamma_robustness_on_the_gpu():
    from alfi_amd import hip
    ctx = hip.Context(0)
    its = {}
    for gamma in (0.0, 1e2, 1e4, 1e6):
        lv, tr = build_sv_hierarchy(TwoDimLidDrivenCavityProblem(2), 2, 2, Re=0, gamma=gamma, advect=False)
        mg = hip.Multigrid(ctx, lv, tr, 3, robust_restriction=True)
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
        its[gamma] = fgmres_solve(A, M, b)
        mg.close()
    assert max(its.values()) <= 12 and its[1e6] - its[1e2] <= 2, its
    assert abs(its[1e4] - run_oracle(1e4, True)) <= 1
    ctx.close()


@pytest.mark.parametrize("case", ["2d-p2", "3d-p3"])
def test_sv_outer_solve_matches_oracle(case):
    """One Newton-step linear solve with the discontinuous P_{k-1} pressure (alfi_saddle_set_mass_inverse: DGMassInv with
    the cell-block mass inverse, solver.py:15-38, 624-629): iteration count and solution against the oracle."""
    from alfi_amd import hip
    from alfi_amd.problem import ThreeDimLidDrivenCavityProblem
    from alfi_amd.sv import build_sv_pressure_coupling
    from oracle import alfi_oracle as O
    if case == "2d-p2":
        lv, tr = build_sv_hierarchy(TwoDimLidDrivenCavityProblem(2), 2, 2, Re=10.0, gamma=1e4)
        k = 6
    else:
        lv, tr = build_sv_hierarchy(ThreeDimLidDrivenCavityProblem(1), 1, 3, Re=10.0, gamma=1e4)
        k = 4
    L = lv[-1]
    B, M, Minv = build_sv_pressure_coupling(L)
    rng = np.random.default_rng(1)
    b = rng.standard_normal(L.n)
    b[L.bc_dofs] = 0
    rhs = np.concatenate([b, np.zeros(B.shape[0])])
    omg = O.build_oracle_mg(lv, tr, k, schoeberl_restriction=False)
    xo, its_o, hist = O.saddle_solve(omg, omg.levels[-1]["A"], B, None, L.nu, L.gamma, rhs, rtol=1e-9, atol=1e-12,
                                     mass_inv=Minv)
    ctx = hip.Context(0)
    mg = hip.Multigrid(ctx, lv, tr, k, robust_restriction=False)
    sad = hip.Saddle(mg, B, None, L.nu, L.gamma, remove_constant_nullspace=True, mass_inv=Minv)
    db, dx = ctx.vec(rhs), ctx.vec(L.n + B.shape[0])
    its, rn = sad.solve(db, dx, 1e-9, 1e-12, 500, 30)
    x = dx.get()
    assert abs(its - its_o) <= 1 and its <= 12, (its, its_o)
    assert rn <= 2e-9 * np.linalg.norm(rhs)
    assert np.abs(x[:L.n] - xo[:L.n]).max() < 1e-6 * np.abs(xo[:L.n]).max()
    sad.close()
    mg.close()
    ctx.close()
