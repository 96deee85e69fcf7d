"""Method of manufactured solutions for the whole solver: the reference's convergence study (examples/mms.py:25-101 with
examples/mmsldc2d/mmsldc2d.py and mmsldc3d/mmsldc3d.py) -- a lid-driven-cavity-like exact solution of the stationary
Navier-Stokes equations (Shih, Tan & Hwang, "Effects of grid staggering on numerical schemes", scaled to [0, 2]^d), the
body force that makes it one, error norms and observed convergence orders.

This is the one check of the generator + solve loop against something that is neither the oracle nor the product: the
continuous problem.  What is restated from the reference: the exact fields (mmsldc2d.py:44-77) and the strong form of the
right-hand side (mmsldc2d.py:79-84), both re-derived symbolically with SymPy instead of UFL.
"""
import numpy as np

from .elements import simplex_quadrature
from .problem import NavierStokesProblem
from .mesh import rectangle_mesh, box_mesh


def _symbolic(dim):
    """lambdified (u, p, f) as functions of the coordinates and Re; char_length 2, char_velocity 1 => nu = 2 / Re."""
    import sympy as sp
    X = sp.symbols("x y z")[:dim]
    Re = sp.Symbol("Re", positive=True)
    xs, ys = sp.symbols("xs ys")                 # coordinates of the unit-square solution
    f = xs ** 4 - 2 * xs ** 3 + xs ** 2
    g = ys ** 4 - ys ** 2
    df, dg = sp.diff(f, xs), sp.diff(g, ys)
    ddg, dddg = sp.diff(g, ys, 2), sp.diff(g, ys, 3)
    F = sp.Rational(1, 5) * xs ** 5 - sp.Rational(1, 2) * xs ** 4 + sp.Rational(1, 3) * xs ** 3
    F2 = f ** 2 / 2
    u0 = 8 * f * dg                                                        # mmsldc2d.py:64
    v0 = -8 * df * g                                                       # :65
    p0 = (8 / Re) * (F * dddg + df * dg) + 64 * F2 * (g * ddg - dg ** 2)   # :66
    half = {xs: X[0] / 2, ys: X[1] / 2}                                    # replace(u, {X: 0.5 * X})   :67-69
    u = [u0.subs(half), v0.subs(half)] + [sp.Integer(0)] * (dim - 2)
    p = p0.subs(half)
    lim = [(c, 0, 2) for c in X]
    p = p - sp.integrate(p, *lim) / 2 ** dim                              # zero mean (the reference subtracts 0.25 assemble(p dx))
    nu = 2 / Re
    grad = [[sp.diff(u[i], X[j]) for j in range(dim)] for i in range(dim)]
    # f1 = -nu div(2 sym grad u) + (grad u) u + grad p          mmsldc2d.py:79-84
    rhs = []
    for i in range(dim):
        visc = sum(sp.diff(grad[i][j] + grad[j][i], X[j]) for j in range(dim))
        adv = sum(grad[i][j] * u[j] for j in range(dim))
        rhs.append(sp.simplify(-nu * visc + adv + sp.diff(p, X[i])))
    args = list(X) + [Re]
    lam = lambda e: sp.lambdify(args, e, "numpy")
    return [lam(c) for c in u], lam(p), [lam(c) for c in rhs], [[lam(c) for c in row] for row in grad]


class _MMSBase(NavierStokesProblem):
    def __init__(self, baseN):
        self.baseN = baseN
        self._u, self._p, self._f, self._gu = _symbolic(self.dim)

    def _cols(self, x, re):
        return [x[:, i] for i in range(self.dim)] + [float(re)]

    def actual_velocity(self, x):
        x = np.atleast_2d(x)
        a = self._cols(x, 1.0)
        return np.stack([np.broadcast_to(np.asarray(c(*a), dtype=np.float64), x.shape[:1]) for c in self._u], axis=1)

    def actual_velocity_gradient(self, x):
        a = self._cols(x, 1.0)
        return np.stack([np.stack([np.broadcast_to(np.asarray(c(*a), dtype=np.float64), x.shape[:1]) for c in row], axis=1)
                         for row in self._gu], axis=1)

    def actual_pressure(self, x, re):
        return np.broadcast_to(np.asarray(self._p(*self._cols(x, re)), dtype=np.float64), x.shape[:1])

    def rhs(self, x, re):
        """Body force of the momentum equation at the points x for Reynolds number re (nu = 2 / re)."""
        a = self._cols(x, re)
        return np.stack([np.broadcast_to(np.asarray(c(*a), dtype=np.float64), x.shape[:1]) for c in self._f], axis=1)

    def driver(self, x):               # Dirichlet data = the exact velocity (zero on the walls, the lid profile on top)
        return self.actual_velocity(x)

    def char_length(self):
        return 2.0

    def relaxation_direction(self):
        return "0+:1-"


class TwoDimLidDrivenCavityMMSProblem(_MMSBase):
    """examples/mmsldc2d/mmsldc2d.py (on the structured triangulation of [0, 2]^2 the file keeps as a comment, :21-23; the
    reference's default is an unstructured gmsh mesh of the same square)."""
    dim = 2

    def mesh(self, distribution_parameters=None):
        return rectangle_mesh(self.baseN, self.baseN, 2.0, 2.0, "left")


class ThreeDimLidDrivenCavityMMSProblem(_MMSBase):
    """examples/mmsldc3d/mmsldc3d.py: the same fields, constant in z, on [0, 2]^3."""
    dim = 3

    def mesh(self, distribution_parameters=None):
        return box_mesh(self.baseN, self.baseN, self.baseN, 2.0, 2.0, 2.0)


def _quadrature(V, n):
    lam, w = simplex_quadrature(V.dim, n)
    m = V.mesh
    xq = np.einsum("qv,cvx->cqx", lam, m.coords[m.cells])           # (cells, nq, dim)
    _, vol = m.cell_geometry()
    return lam, w, xq, vol


def load_vector(V, f, n=6):
    """int f . phi_i for the vector space V (interleaved dofs); f(points (N, dim)) -> (N, dim); rule exact for degree 2n-1."""
    lam, w, xq, vol = _quadrature(V, n)
    phi, _ = V.element.tabulate(lam)                                  # (nq, nloc)
    nc, nq, d = xq.shape
    fq = f(xq.reshape(-1, d)).reshape(nc, nq, d)
    loc = np.einsum("q,c,cqx,qa->cax", w, vol, fq, phi)              # (cells, nloc, dim)
    out = np.zeros((V.num_nodes, d))
    np.add.at(out, V.cell_nodes, loc)
    return out.ravel()


def errors(V, u_h, problem, re, p_cell=None, p_eval=None, n=6):
    """L2 error of the velocity, of its gradient, of the pressure (mean-free), and |div u_h|_{L2}.
    p_cell: one value per cell (P0);  p_eval(lam, cells) -> (cells, nq) for other pressure spaces."""
    lam, w, xq, vol = _quadrature(V, n)
    phi, dphi = V.element.tabulate(lam)
    g, _ = V.mesh.cell_geometry()                                      # (cells, dim+1, dim): grad lambda_i
    nc, nq, d = xq.shape
    uc = u_h.reshape(-1, d)[V.cell_nodes]                              # (cells, nloc, dim)
    uq = np.einsum("qa,cax->cqx", phi, uc)
    gphi = np.einsum("qai,cix->cqax", dphi, g)                         # physical gradients of the basis
    guq = np.einsum("cqay,cax->cqxy", gphi, uc)                        # d_y u_x
    pts = xq.reshape(-1, d)
    ue = problem.actual_velocity(pts).reshape(nc, nq, d)
    gue = problem.actual_velocity_gradient(pts).reshape(nc, nq, d, d)
    wv = w[None, :] * vol[:, None]
    out = {"velocity": float(np.sqrt((wv * ((uq - ue) ** 2).sum(axis=2)).sum())),
           "velocitygrad": float(np.sqrt((wv * ((guq - gue) ** 2).sum(axis=(2, 3))).sum())),
           "divergence": float(np.sqrt((wv * np.trace(guq, axis1=2, axis2=3) ** 2).sum()))}
    if p_cell is not None or p_eval is not None:
        ph = np.broadcast_to(p_cell[:, None], (nc, nq)) if p_cell is not None else p_eval(lam, np.arange(nc))
        pe = problem.actual_pressure(pts, re).reshape(nc, nq)
        area = float(vol.sum())
        ph = ph - (wv * ph).sum() / area
        pe = pe - (wv * pe).sum() / area
        out["pressure"] = float(np.sqrt((wv * (ph - pe) ** 2).sum()))
    return out


def convergence_orders(x):
    """examples/mms.py:11."""
    x = np.asarray(x, dtype=np.float64)
    return np.log2(x[:-1] / x[1:])
