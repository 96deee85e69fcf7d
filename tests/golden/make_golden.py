"""Generates tests/golden/*.npz with the oracle (there are no reference fixtures to take: SURVEY.md section 4).
Run from the repo root: python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from alfi_amd.problem import TwoDimLidDrivenCavityProblem, ThreeDimLidDrivenCavityProblem, build_hierarchy  # noqa: E402
from alfi_amd.problem import build_pressure_coupling  # noqa: E402
from alfi_amd.relaxation import OrderedRelaxation, Options  # noqa: E402
from oracle import alfi_oracle as O  # noqa: E402

CASES = {"ldc2d_p2_N4": (lambda: TwoDimLidDrivenCavityProblem(2), 2, 1, 10.0, 6),
         "ldc3d_p1fb_N2": (lambda: ThreeDimLidDrivenCavityProblem(1), 1, 1, 100.0, 4),
         "ldc3d_p2fb_N2": (lambda: ThreeDimLidDrivenCavityProblem(1), 2, 1, 100.0, 4)}


def sweep_order(L):
    """Iteration set of the Star constructor with the problems' relaxation_direction "0+:1-" (relaxation.py:139-150)."""
    orl = OrderedRelaxation()
    orl.name = "Star"
    orl.opts = Options("", {"pc_patch_construction_Star_sort_order": "0+:1-"})
    return orl.iteration_order(L.V.mesh.coords[L.patch_seeds])


def make(name):
    mk, k, nref, Re, ks = CASES[name]
    lv, tr = build_hierarchy(mk(), nref, k, Re=Re)
    mg = O.build_oracle_mg(lv, tr, ks, schoeberl_restriction=True)
    L = lv[-1]
    rng = np.random.default_rng(42)
    x = rng.standard_normal(L.n)
    b = rng.standard_normal(L.n)
    b[L.bc_dofs] = 0
    uc = rng.standard_normal(lv[0].n)
    uc[lv[0].bc_dofs] = 0
    out = dict(x=x, b=b, uc=uc, patch_ptr=L.patch_ptr, patch_dofs=L.patch_dofs, n=L.n,
               A_x=mg.levels[-1]["A"] @ x, patch_apply_x=mg.levels[-1]["smoother"].apply(x),
               smooth_b=mg.smooth(len(lv) - 1, b, np.zeros(L.n)),
               prolong_uc=mg.prolong(len(lv) - 1, uc), restrict_b=mg.restrict(len(lv) - 1, x),
               vcycle_b=mg.vcycle(len(lv) - 1, b, np.zeros(L.n)), fcycle_b=mg.fcycle(b),
               inv_patch0=mg.levels[-1]["smoother"].inv[0])
    # multiplicative symmetrised sweep in the problems' relaxation order; inject; one outer (saddle-point) solve
    order = sweep_order(L)
    sm = O.PatchSmoother(mg.levels[-1]["A"], L.patch_ptr, L.patch_dofs, L.bc_dofs, "multiplicative", order, True)
    B, vol = build_pressure_coupling(L)
    mgp = O.build_oracle_mg(lv, tr, ks)
    rhs = np.concatenate([b, np.zeros(B.shape[0])])
    xs, its, hist = O.saddle_solve(mgp, mg.levels[-1]["A"], B, vol, L.nu, L.gamma, rhs, rtol=1e-9, atol=1e-10)
    out.update(mult_order=order, mult_apply_x=sm.apply(x), inject_x=x.reshape(-1, L.bs)[tr[-1].inject_map].ravel(),
               saddle_x=xs, saddle_its=its, saddle_hist=np.array(hist))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), name + ".npz"), **out)


# Scott-Vogelius side (alfi_amd/sv.py): name -> (problem, nref, element degree, Re, smoothing steps)
SV_CASES = {"sv2d_p2_N2": (lambda: TwoDimLidDrivenCavityProblem(2), 2, 2, 10.0, 6),
            "sv3d_p3_N1": (lambda: ThreeDimLidDrivenCavityProblem(1), 1, 3, 100.0, 4)}


def sv_hierarchy(name):
    from alfi_amd.sv import build_sv_hierarchy
    mk, nref, k, Re, ks = SV_CASES[name]
    lv, tr = build_sv_hierarchy(mk(), nref, k, Re=Re)
    return lv, tr, ks


def make_sv(name):
    from alfi_amd.sv import build_sv_pressure_coupling
    lv, tr, ks = sv_hierarchy(name)
    mg = O.build_oracle_mg(lv, tr, ks, schoeberl_restriction=True)
    L = lv[-1]
    top = len(lv) - 1
    rng = np.random.default_rng(7)
    x = rng.standard_normal(L.n)
    b = rng.standard_normal(L.n)
    b[L.bc_dofs] = 0
    uc = rng.standard_normal(lv[-2].n)
    uc[lv[-2].bc_dofs] = 0
    B, M, Minv = build_sv_pressure_coupling(L)
    mgp = O.build_oracle_mg(lv, tr, ks)
    rhs = np.concatenate([b, np.zeros(B.shape[0])])
    xs, its, hist = O.saddle_solve(mgp, mg.levels[-1]["A"], B, None, L.nu, L.gamma, rhs, rtol=1e-9, atol=1e-12, mass_inv=Minv)
    out = dict(x=x, b=b, uc=uc, n=L.n, patch_ptr=L.patch_ptr, patch_dofs=L.patch_dofs, blk_dofs=tr[-1].blk_dofs,
               A_x=mg.levels[-1]["A"] @ x, patch_apply_x=mg.levels[-1]["smoother"].apply(x),
               prolong_uc=mg.prolong(top, uc), restrict_x=mg.restrict(top, x), vcycle_b=mg.vcycle(top, b, np.zeros(L.n)),
               fcycle_b=mg.fcycle(b), inject_x=tr[-1].inject_matrix @ x.reshape(-1, L.bs), saddle_x=xs, saddle_its=its,
               B_x=B @ x, Minv_diag=Minv.diagonal())
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), name + ".npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or list(CASES) + list(SV_CASES)
    for name in which:
        (make if name in CASES else make_sv)(name)
        print("wrote", name)
