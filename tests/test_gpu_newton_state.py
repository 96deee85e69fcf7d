"""The Newton loop with its state resident on the device (alfi_amd.nssolver.HipNavierStokesSolver._solve_on_device): what
crosses PCIe per Newton step is counted by the library itself (alfi_transfer_stats: every hipMemcpy it makes) and must be
scalars only -- norms, iteration counts, probe results --, as the reference keeps z as a Function on its ranks and never
gathers it (alfi/solver.py:245-273).  -m gpu."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from alfi_amd.problem import TwoDimLidDrivenCavityProblem, ThreeDimLidDrivenCavityProblem


@pytest.mark.parametrize("mk,k,nref,supg", [(lambda: TwoDimLidDrivenCavityProblem(8), 2, 2, False),
                                            (lambda: ThreeDimLidDrivenCavityProblem(2), 2, 1, False),
                                            (lambda: ThreeDimLidDrivenCavityProblem(2), 1, 2, True)],
                         ids=["ldc2d", "ldc3d-P2FB", "ldc3d-P1FB-supg"])
def test_a_newton_step_moves_only_scalars_across_pcie(mk, k, nref, supg):
    from alfi_amd.nssolver import HipNavierStokesSolver
    kw = dict(stabilisation_type="supg", stabilisation_weight=0.05) if supg else {}
    s = HipNavierStokesSolver(mk(), nref, k, device_assembly=True, **kw)
    s.solve(10)                                   # (first solve: uploads the initial state once, allocates scratch)
    s.ctx.transfer_stats(reset=True)
    _, info = s.solve(100)
    h2d, d2h = s.ctx.transfer_stats()
    steps = info["nonlinear_iter"]
    assert steps >= 2 and info["converged"]
    # per Newton step: a few dozen 4- and 8-byte reads (Krylov residual norms, patch probe results, norms of F, dz, z) and no
    # vector; the state itself (tens of kilobytes here, 94 MB at config-4 size) never moves
    n_state = (s.n_u + s.n_p) * 8
    assert h2d / steps <= 1024 and d2h / steps <= 1024, (h2d, d2h, steps, n_state)
    assert n_state > 20 * 1024
    # the state is fetched when somebody asks for it -- once
    u = s.u
    got = s.ctx.transfer_stats()[1] - d2h            # (+ the 4-byte device error word every synchronising read checks)
    assert n_state <= got <= n_state + 16 and u.shape == (s.n_u,)
    _ = s.p
    assert s.ctx.transfer_stats()[1] - d2h == got
    s.close()


def test_device_resident_newton_equals_the_host_state_loop():
    """Same Newton / Krylov counts and the same solution as the loop that keeps the state on the host and re-assembles the
    operators there (device_assembly=False: the round-2 path)."""
    from alfi_amd.nssolver import HipNavierStokesSolver, run_solver
    out = {}
    for dev in (True, False):
        s = HipNavierStokesSolver(ThreeDimLidDrivenCavityProblem(2), 1, 2, device_assembly=dev)
        res = run_solver(s, [10, 100])
        out[dev] = (s.u.copy(), s.p.copy(), [(res[r]["nonlinear_iter"], res[r]["linear_iter"], res[r]["converged"]) for r in (10, 100)],
                    [res[r]["residual_history"] for r in (10, 100)])
        s.close()
    assert out[True][2] == out[False][2] and all(c for _, _, c in out[True][2])
    assert np.abs(out[True][0] - out[False][0]).max() <= 1e-8 * np.abs(out[False][0]).max()
    assert np.abs(out[True][1] - out[False][1]).max() <= 1e-7 * np.abs(out[False][1]).max()
    for a, b in zip(out[True][3], out[False][3]):
        assert len(a) == len(b) and abs(a[0] - b[0]) <= 1e-10 * b[0]
