"""Condensed patch factors (alfi_patches_set_groups): the exact block factorisation of the macro-star patch inverses --
interiors of the macro cells + skeleton -- against the dense inverses and the oracle.  -m gpu.

Reference: macro-star patches of the Scott-Vogelius solver (alfi/relaxation.py:163-177, alfi/solver.py:339-342) on
Alfeld-split meshes (alfi/bary.py), where the reference keeps sparse LU factors of the patch matrices
(alfi/solver.py:655-659) instead of dense inverses."""
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


@pytest.mark.parametrize("case", ["2d-P2", "3d-P2", "3d-P3"])
def test_condensed_apply_equals_dense_inverse_apply(ctx, case):
    from alfi_amd import hip
    from alfi_amd.problem import TwoDimLidDrivenCavityProblem, ThreeDimLidDrivenCavityProblem
    from alfi_amd.sv import build_sv_hierarchy
    from oracle import alfi_oracle as O
    prob, nref, k = {"2d-P2": (TwoDimLidDrivenCavityProblem(2), 2, 2), "3d-P2": (ThreeDimLidDrivenCavityProblem(1), 1, 2),
                     "3d-P3": (ThreeDimLidDrivenCavityProblem(1), 1, 3)}[case]
    lv, _ = build_sv_hierarchy(prob, nref, k, Re=100.0)
    L = lv[-1]
    assert (L.patch_groups >= 0).any() and (L.patch_groups < 0).any()
    x = np.random.default_rng(0).standard_normal(L.n)
    out = {}
    for mode in ("dense", "condensed"):
        dl = hip.Level(ctx, L.A, L.bc_dofs)
        dl.set_patches(L.patch_ptr, L.patch_dofs)
        if mode == "condensed":
            dl.set_patch_groups(L.patch_groups)
        dl.factor()
        worst, flagged, _, _ = dl.patch_check()
        assert 0.0 <= worst < 1e-7 and flagged == 0
        dx, dy = ctx.vec(x), ctx.vec(L.n)
        dl.patch_apply(dx, dy)
        out[mode] = (dy.get(), dl.factor_bytes())
        dl.patch_apply(dx, dy)
        assert np.array_equal(dy.get(), out[mode][0])               # deterministic
        if mode == "condensed":                                      # and back to dense inverses on the same level
            dl.set_patch_groups(None)
            dl.factor()
            dl.patch_apply(dx, dy)
            assert np.array_equal(dy.get(), out["dense"][0]) and dl.factor_bytes() == out["dense"][1]
        dl.close()
    ref = O.PatchSmoother(L.A.to_scipy().tocsr(), L.patch_ptr, L.patch_dofs, L.bc_dofs).apply(x)
    assert relerr(out["dense"][0], ref) < 1e-7
    assert relerr(out["condensed"][0], ref) < 1e-7
    assert relerr(out["condensed"][0], out["dense"][0]) < 1e-8
    n2 = float((np.diff(L.patch_ptr).astype(np.float64) ** 2).sum())
    print("%s: %d patches up to %d dofs, dense %.2f MB, condensed %.2f MB (%.1f x)"
          % (case, len(L.patch_ptr) - 1, np.diff(L.patch_ptr).max(), out["dense"][1] / 1e6, out["condensed"][1] / 1e6,
             out["dense"][1] / out["condensed"][1]))
    assert out["condensed"][1] < 8 * n2
    if case == "3d-P3":
        assert out["dense"][1] > 3 * out["condensed"][1]


def test_groups_that_are_coupled_are_refused(ctx):
    from alfi_amd import hip
    from alfi_amd.problem import TwoDimLidDrivenCavityProblem
    from alfi_amd.sv import build_sv_hierarchy
    lv, _ = build_sv_hierarchy(TwoDimLidDrivenCavityProblem(2), 1, 2, Re=10.0)
    L = lv[-1]
    dl = hip.Level(ctx, L.A, L.bc_dofs)
    dl.set_patches(L.patch_ptr, L.patch_dofs)
    bad = np.arange(len(L.patch_dofs), dtype=np.int32) // L.bs        # every node its own group: neighbours are coupled
    with pytest.raises(hip.AlfiHipError, match="coupled"):
        dl.set_patch_groups(bad)
    dl.set_patch_groups(None)                                          # back to dense
    dl.factor()
    dl.close()


def test_cycles_with_condensed_factors_match_the_oracle(ctx):
    """The whole PCMG cycle of the 3-D [P3]^3 Scott-Vogelius hierarchy (what hip.Multigrid builds by default: condensed
    macro-star factors) against the oracle."""
    from alfi_amd import hip
    from alfi_amd.problem import ThreeDimLidDrivenCavityProblem
    from alfi_amd.sv import build_sv_hierarchy
    from oracle import alfi_oracle as O
    lv, tr = build_sv_hierarchy(ThreeDimLidDrivenCavityProblem(1), 1, 3, Re=100.0)
    mg = hip.Multigrid(ctx, lv, tr, 3, robust_restriction=True)
    assert mg.levels[-1].factor_bytes() < 0.5 * 8 * float((np.diff(lv[-1].patch_ptr).astype(np.float64) ** 2).sum())
    omg = O.build_oracle_mg(lv, tr, 3, schoeberl_restriction=True)
    L = lv[-1]
    b = np.random.default_rng(0).standard_normal(L.n)
    b[L.bc_dofs] = 0
    db, dx = ctx.vec(b), ctx.vec(L.n)
    mg.vcycle(db, dx)
    assert relerr(dx.get(), omg.vcycle(1, b, np.zeros(L.n))) < 1e-5
    mg.fcycle(db, dx)
    assert relerr(dx.get(), omg.fcycle(b)) < 1e-5
    mg.close()


def test_flagged_condensed_factors_are_repaired_in_place():
    """A condensed factor that fails the residual probe of alfi_patches_factor keeps its condensed storage: its Schur
    complement is formed again and inverted by LU with partial pivoting (round 2 fell back to dense inverses for the whole
    level).  Forced here for every patch by a probe tolerance no inverse reaches (own process: the library reads its
    switches once)."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ALFI_PATCH_CHECK_TOL="1e-14", ALFI_PATCH_CHECK_FAIL="1e-6")
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_cond_repair_worker.py")], env=env, cwd=root,
                         capture_output=True, text=True, timeout=900)
    m = re.search(r"CHECK (\d+) (\d+) (\S+) (\S+) BYTES (\d+) (\d+) RELERR (\S+)", out.stdout)
    assert out.returncode == 0 and m, out.stdout[-2000:] + out.stderr[-3000:]
    flagged, repaired = int(m.group(1)), int(m.group(2))
    worst, after, err = float(m.group(3)), float(m.group(4)), float(m.group(7))
    assert flagged > 0                                     # every patch went through the repair ...
    assert after < 1e-7 and err < 1e-7                     # ... and comes out as accurate as the fast factorisation
    assert int(m.group(5)) < int(m.group(6)) / 2           # still condensed: far below the 8 sum n_p^2 bytes of dense inverses


def test_zero_pivot_inside_a_group_is_repaired_in_place(ctx):
    """a zero (or tiny) leading pivot inside ONE macro-cell group used to fail the unpivoted group
    elimination, and hip.factor_with_fallback then turned the whole level into dense inverses (6.8 x the memory: 265 GB at
    config 5L).  The reference's sparse patch LU pivots (alfi/solver.py:655-659).  Here the [P3]^3 operator keeps its sparsity
    but gets values no unpivoted elimination survives -- the (0, 0) entry of every diagonal 3 x 3 block is zero, so the first
    pivot of every group and of every Schur complement is zero while the patch matrices stay well conditioned -- and every
    patch comes back from the repair (cond_group_pivot_kernel + pivoted LU of the Schur complement) with its condensed
    storage: factor_bytes() unchanged, probe < 1e-7, apply equal to dense solves with numpy."""
    from alfi_amd import hip
    from alfi_amd.problem import ThreeDimLidDrivenCavityProblem, BSR
    from alfi_amd.sv import build_sv_hierarchy
    lv, _ = build_sv_hierarchy(ThreeDimLidDrivenCavityProblem(1), 1, 3, Re=100.0)
    L = lv[-1]
    A0 = L.A
    rng = np.random.default_rng(0)
    vals = 0.02 * rng.standard_normal(A0.vals.shape)
    rows = np.repeat(np.arange(A0.nbrows), np.diff(A0.rowptr))
    diag = rows == A0.colidx
    base = np.array([[0.0, 4.0, 0.0], [4.0, 0.0, 1.0], [0.0, 1.0, 5.0]])
    vals[diag] = base + 0.02 * rng.standard_normal((int(diag.sum()), 3, 3))
    vals[diag, 0, 0] = 0.0
    A = BSR(A0.nbrows, A0.nbcols, 3, A0.rowptr, A0.colidx, vals)
    # reference: healthy values, condensed storage
    dl0 = hip.Level(ctx, A0, L.bc_dofs)
    dl0.set_patches(L.patch_ptr, L.patch_dofs)
    dl0.set_patch_groups(L.patch_groups)
    dl0.factor()
    healthy_bytes = dl0.factor_bytes()
    dl0.close()
    dl = hip.Level(ctx, A, np.zeros(0, dtype=np.int32))
    dl.set_patches(L.patch_ptr, L.patch_dofs)
    dl.set_patch_groups(L.patch_groups)
    assert dl.factor_with_fallback() is False                 # repaired in place: no dense fallback
    worst, flagged, repaired, after = dl.patch_check()
    npatch = len(L.patch_ptr) - 1
    assert flagged == npatch == repaired and not (worst < 1e-6), (worst, flagged, repaired, after)
    assert after < 1e-7, after
    assert dl.factor_bytes() == healthy_bytes                 # still condensed
    S = A.to_scipy().tocsr()
    x = rng.standard_normal(L.n)
    dx, dy = ctx.vec(x), ctx.vec(L.n)
    dl.patch_apply(dx, dy)
    y = np.zeros(L.n)
    for p in range(npatch):
        dofs = L.patch_dofs[L.patch_ptr[p]:L.patch_ptr[p + 1]]
        y[dofs] += np.linalg.solve(S[dofs][:, dofs].toarray(), x[dofs])
    assert np.abs(dy.get() - y).max() < 1e-9 * np.abs(y).max()
    dl.close()
