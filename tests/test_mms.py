"""Manufactured solutions (examples/mms.py, examples/mmsldc2d): the exact fields and the quadrature helpers, on the CPU.
The convergence study itself needs the GPU solve loop: tests/test_gpu_mms.py."""
import numpy as np


def test_exact_solution_satisfies_the_strong_equations():
    """The symbolic fields themselves (no GPU involved): divergence-free, zero on the walls, unit lid speed at the top
    centre, and -nu div(2 sym grad u) + (grad u) u + grad p = f by central differences."""
    from alfi_amd.mms import TwoDimLidDrivenCavityMMSProblem
    pr = TwoDimLidDrivenCavityMMSProblem(4)
    rng = np.random.default_rng(0)
    x = rng.uniform(0.1, 1.9, size=(20, 2))
    h = 1e-4
    E = np.eye(2)
    d = lambda fun, i: (fun(x + h * E[i]) - fun(x - h * E[i])) / (2 * h)
    u = pr.actual_velocity
    assert np.abs(np.trace(pr.actual_velocity_gradient(x), axis1=1, axis2=2)).max() < 1e-13
    for i in range(2):
        for j in range(2):
            assert np.allclose(pr.actual_velocity_gradient(x)[:, i, j], d(u, j)[:, i], atol=1e-6)
    re = 10.0
    nu = 2.0 / re
    lap = sum((u(x + h * E[i]) - 2 * u(x) + u(x - h * E[i])) / h ** 2 for i in range(2))       # div(2 sym grad u) = lap u
    gu = pr.actual_velocity_gradient(x)
    adv = np.einsum("nij,nj->ni", gu, u(x))
    gp = np.stack([d(lambda y: pr.actual_pressure(y, re), i) for i in range(2)], axis=1)
    assert np.abs(-nu * lap + adv + gp - pr.rhs(x, re)).max() < 1e-5
    wall = np.array([[0.0, 0.7], [2.0, 1.3], [0.9, 0.0]])
    assert np.abs(u(wall)).max() < 1e-14
    assert np.allclose(u(np.array([[1.0, 2.0]])), [[1.0, 0.0]])




def test_load_vector_and_error_norms():
    """int f . phi_i summed over the nodes is int f (the nodal basis is a partition of unity); the error functional of the
    interpolated exact velocity decays with the interpolation order of [P2]^2."""
    from alfi_amd.mms import TwoDimLidDrivenCavityMMSProblem, errors, load_vector
    from alfi_amd.problem import build_hierarchy
    pr = TwoDimLidDrivenCavityMMSProblem(4)
    lv, _ = build_hierarchy(pr, 2, 2, Re=10.0, patches=False)
    V = lv[-1].V
    ld = load_vector(V, lambda y: np.stack([1.0 + y[:, 0], y[:, 0] * y[:, 1]], axis=1)).reshape(-1, 2)
    assert np.allclose(ld.sum(axis=0), [4.0 + 4.0, 4.0], rtol=1e-12)        # int over [0,2]^2 of (1 + x, x y)
    e = [errors(L.V, pr.actual_velocity(L.V.node_coords).ravel(), pr, 10.0)["velocity"] for L in lv]
    orders = np.log2(np.array(e[:-1]) / np.array(e[1:]))
    assert (orders > 2.7).all(), (e, orders)
