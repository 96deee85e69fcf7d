"""The C/OpenMP oracle port (cpu_baseline) against the NumPy oracle: same algorithm, so tight tolerances."""
import numpy as np
import pytest

from alfi_amd.problem import TwoDimLidDrivenCavityProblem, ThreeDimLidDrivenCavityProblem, build_hierarchy
from oracle import alfi_oracle as O
from oracle import c_oracle as C


@pytest.mark.parametrize("mk,k,nref,Re", [(lambda: TwoDimLidDrivenCavityProblem(4), 2, 2, 100.0),
                                          (lambda: ThreeDimLidDrivenCavityProblem(2), 1, 1, 100.0),
                                          (lambda: ThreeDimLidDrivenCavityProblem(2), 2, 1, 500.0)])
@pytest.mark.parametrize("robust", [False, True])
def test_c_port_matches_numpy_oracle(mk, k, nref, Re, robust):
    lv, tr = build_hierarchy(mk(), nref, k, Re=Re)
    omg = O.build_oracle_mg(lv, tr, 3, schoeberl_restriction=robust)
    cmg = C.CMultigrid(lv, tr, 3, robust_restriction=robust)
    L = lv[-1]
    rng = np.random.default_rng(0)
    x = rng.standard_normal(L.n)
    ref = omg.levels[-1]["smoother"].apply(x)
    assert np.abs(cmg.levels[-1].patch_apply(x) - ref).max() < 1e-8 * np.abs(ref).max()
    b = rng.standard_normal(L.n)
    b[L.bc_dofs] = 0
    ref = omg.vcycle(len(lv) - 1, b, np.zeros(L.n))
    got = cmg.vcycle(len(lv) - 1, b, np.zeros(L.n))
    assert np.abs(got - ref).max() < 1e-5 * np.abs(ref).max()
    xc = rng.standard_normal(lv[-2].n)
    xc[lv[-2].bc_dofs] = 0
    ref = omg.prolong(len(lv) - 1, xc)
    assert np.abs(cmg.transfers[-1].prolong(xc) - ref).max() < 1e-9 * np.abs(ref).max()


def test_c_port_on_a_scott_vogelius_hierarchy():
    """Macro stars of up to 513 dofs and macro-cell blocks of 123 (beyond the sizes of the PkP0 configurations): the
    port is the cpu_baseline of ``bench.py --config cfg5``."""
    from alfi_amd.sv import build_sv_hierarchy
    lv, tr = build_sv_hierarchy(ThreeDimLidDrivenCavityProblem(1), 1, 2, Re=100.0, gamma=1e2)
    assert np.diff(lv[1].patch_ptr).max() > 256 and tr[0].blk_dofs.shape[1] == 123
    omg = O.build_oracle_mg(lv, tr, 3, schoeberl_restriction=True)
    cmg = C.CMultigrid(lv, tr, 3, robust_restriction=True)
    L = lv[-1]
    rng = np.random.default_rng(0)
    b = rng.standard_normal(L.n)
    b[L.bc_dofs] = 0
    ref = omg.vcycle(1, b, np.zeros(L.n))
    got = cmg.vcycle(1, b, np.zeros(L.n))
    assert np.abs(got - ref).max() < 1e-6 * np.abs(ref).max()
