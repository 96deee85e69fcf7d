"""Nodal velocity elements of the reference's PkP0 discretisation, expressed in barycentric coordinates.

alfi/solver.py:574-586 picks ``VectorElement(NodalEnrichedElement(P_k, FacetBubble))`` when k < tdim and plain
``VectorElement(P_k)`` otherwise, i.e. [P2]^2 in 2-D (configs 1-2), [P1+FB]^3 (config 3) and [P2+FB]^3 (config 4)
in 3-D.  All dofs are point evaluations (vertices, edge midpoints, face barycentres; SURVEY.md Appendix B, evidence:
the change of basis hard-coded in alfi/bubble.py:64-80), so the nodal basis is

    phi~_a = phi_a - sum_m phi_a(barycentre of face m) * beta_m,      beta_m = 27 * prod_{j != m} lambda_j,

with phi_a the plain Lagrange basis and beta_m the facet bubbles.  Local node order: vertices, (edges), (faces), face
i opposite vertex i (bubble.py:43-44, 73-76).

Only reference-cell quantities live here; they are consumed by the assembly helper (csrc/host_assemble.cpp) and by
the transfer-matrix construction in ``fespace.py``.
"""
import numpy as np
from .mesh import TRI_EDGES, TET_EDGES, TET_FACES


def simplex_quadrature(dim, n):
    """Collapsed Gauss-Jacobi rule with n^dim points, exact for degree 2n-1.  Returns barycentric points
    (npts, dim+1) and weights summing to 1 (i.e. averages over the reference cell)."""
    from scipy.special import roots_jacobi

    def rule(alpha):
        x, w = roots_jacobi(n, alpha, 0)
        return 0.5 * (x + 1.0), w / 2.0 ** (alpha + 1)

    if dim == 2:
        u, wu = rule(1)
        v, wv = rule(0)
        U, V = np.meshgrid(u, v, indexing="ij")
        W = np.outer(wu, wv)
        x, y = U.ravel(), (V * (1 - U)).ravel()
        lam = np.stack([1 - x - y, x, y], axis=1)
    else:
        u, wu = rule(2)
        v, wv = rule(1)
        t, wt = rule(0)
        U, V, T = np.meshgrid(u, v, t, indexing="ij")
        W = wu[:, None, None] * wv[None, :, None] * wt[None, None, :]
        x, y, z = U.ravel(), (V * (1 - U)).ravel(), (T * (1 - U) * (1 - V)).ravel()
        lam = np.stack([1 - x - y - z, x, y, z], axis=1)
    w = W.ravel()
    return lam, w / w.sum()


class NodalElement(object):
    """Scalar nodal element; the velocity space is its dim-fold vector version (node-major, component-minor,
    bubble.py:86)."""

    def __init__(self, dim, degree, bubble):
        assert dim in (2, 3) and degree in (1, 2, 3)
        if bubble:
            assert dim == 3, "FacetBubble enrichment is only needed for k < tdim in 3-D here"
        if degree == 3:
            # the velocity element of the reference's 3-D Scott-Vogelius pair (solver.py:625-630, config 5): 4 vertex,
            # 12 edge (two per edge, at 1/3 and 2/3) and 4 face nodes; no cell-interior node in 3-D
            assert dim == 3 and not bubble, "P3 is provided for tetrahedra (the 2-D pairs of the reference use k = 2)"
        self.dim, self.degree, self.bubble = dim, degree, bubble
        self.name = "P%d%s" % (degree, "+FB" if bubble else "")
        nv = dim + 1
        self.local_edges = TRI_EDGES if dim == 2 else TET_EDGES
        self.nodes_per_edge = {1: 0, 2: 1, 3: 2}[degree]
        ent = [(0, i, 0) for i in range(nv)]
        bary = [np.eye(nv)[i] for i in range(nv)]
        for j, (a, b) in enumerate(self.local_edges):
            for sub in range(self.nodes_per_edge):
                ent.append((1, j, sub))             # sub counts from local vertex a towards b
                p = np.zeros(nv)
                t = (sub + 1.0) / (self.nodes_per_edge + 1.0)
                p[a], p[b] = 1.0 - t, t
                bary.append(p)
        if bubble or degree == 3:
            for i in range(4):
                ent.append((2, i, 0))
                p = np.full(4, 1.0 / 3.0)
                p[i] = 0.0
                bary.append(p)
        self.entity_nodes = ent                     # (entity dim, local entity number, sub-index on the entity) per node
        self.node_bary = np.array(bary)             # (nloc, dim+1)
        self.nloc = len(ent)
        self.has_edge_nodes = degree >= 2
        self.has_face_nodes = bubble or degree == 3
        if bubble:
            fb = self.node_bary[-4:]
            phi, _ = self._primal(fb)               # (4 faces, nprimal)
            self._bub_coef = phi.T.copy()           # coef[a, m] = phi_a(barycentre of face m)

    # plain Lagrange part --------------------------------------------------------------------------------------
    def _primal(self, lam):
        npts, nv = lam.shape
        if self.degree == 3:
            return self._p3(lam)
        if self.degree == 1:
            phi = lam.copy()
            dphi = np.broadcast_to(np.eye(nv), (npts, nv, nv)).copy()
            return phi, dphi
        ne = self.local_edges.shape[0]
        phi = np.empty((npts, nv + ne))
        dphi = np.zeros((npts, nv + ne, nv))
        for i in range(nv):
            phi[:, i] = lam[:, i] * (2 * lam[:, i] - 1)
            dphi[:, i, i] = 4 * lam[:, i] - 1
        for j, (a, b) in enumerate(self.local_edges):
            phi[:, nv + j] = 4 * lam[:, a] * lam[:, b]
            dphi[:, nv + j, a] = 4 * lam[:, b]
            dphi[:, nv + j, b] = 4 * lam[:, a]
        return phi, dphi

    def _p3(self, lam):
        """Cubic Lagrange basis on equispaced nodes: vertices 1/2 l (3l - 1)(3l - 2); edge (a, b) node nearer a:
        9/2 la lb (3 la - 1), nearer b: 9/2 la lb (3 lb - 1); face (i, j, k): 27 li lj lk."""
        npts, nv = lam.shape
        ne = self.local_edges.shape[0]
        n = nv + 2 * ne + 4
        phi = np.empty((npts, n))
        dphi = np.zeros((npts, n, nv))
        for i in range(nv):
            l = lam[:, i]
            phi[:, i] = 0.5 * l * (3 * l - 1) * (3 * l - 2)
            dphi[:, i, i] = 0.5 * (27 * l * l - 18 * l + 2)
        for j, (a, b) in enumerate(self.local_edges):
            la, lb = lam[:, a], lam[:, b]
            k0, k1 = nv + 2 * j, nv + 2 * j + 1
            phi[:, k0] = 4.5 * la * lb * (3 * la - 1)
            dphi[:, k0, a] = 4.5 * lb * (6 * la - 1)
            dphi[:, k0, b] = 4.5 * la * (3 * la - 1)
            phi[:, k1] = 4.5 * la * lb * (3 * lb - 1)
            dphi[:, k1, a] = 4.5 * lb * (3 * lb - 1)
            dphi[:, k1, b] = 4.5 * la * (6 * lb - 1)
        beta, dbeta = self._bubbles(lam)
        phi[:, nv + 2 * ne:] = beta
        dphi[:, nv + 2 * ne:, :] = dbeta
        return phi, dphi

    @staticmethod
    def _bubbles(lam):
        npts = lam.shape[0]
        beta = np.empty((npts, 4))
        dbeta = np.zeros((npts, 4, 4))
        for m in range(4):
            j, k, l = TET_FACES[m]
            beta[:, m] = 27 * lam[:, j] * lam[:, k] * lam[:, l]
            dbeta[:, m, j] = 27 * lam[:, k] * lam[:, l]
            dbeta[:, m, k] = 27 * lam[:, j] * lam[:, l]
            dbeta[:, m, l] = 27 * lam[:, j] * lam[:, k]
        return beta, dbeta

    def tabulate(self, lam):
        """phi (npts, nloc) and dphi (npts, nloc, dim+1) = partial derivatives w.r.t. the barycentric coordinates
        (treated as independent variables; physical gradient = sum_i dphi[..., i] * grad(lambda_i))."""
        lam = np.atleast_2d(np.asarray(lam, dtype=np.float64))
        phi, dphi = self._primal(lam)
        if not self.bubble:
            return phi, dphi
        beta, dbeta = self._bubbles(lam)
        phi = phi - beta @ self._bub_coef.T
        dphi = dphi - np.einsum("pmi,am->pai", dbeta, self._bub_coef)
        return np.concatenate([phi, beta], axis=1), np.concatenate([dphi, dbeta], axis=1)

    def tabulate_hessian(self, lam):
        """d2phi (npts, nloc, dim+1, dim+1): second partial derivatives w.r.t. the barycentric coordinates (independent
        variables, as in ``tabulate``).  The first derivatives are polynomials of degree <= 2 in every coordinate, so
        their central differences are exact."""
        lam = np.atleast_2d(np.asarray(lam, dtype=np.float64))
        nv = lam.shape[1]
        step = 0.5
        out = np.empty((lam.shape[0], self.nloc, nv, nv))
        for k in range(nv):
            e = np.zeros(nv)
            e[k] = step
            out[:, :, :, k] = (self.tabulate(lam + e)[1] - self.tabulate(lam - e)[1]) / (2 * step)
        return out

    # reference tensors (averages over the reference cell: multiply by the cell volume) -----------------------------
    def reference_tensors(self):
        if hasattr(self, "_tensors"):
            return self._tensors
        lam, w = simplex_quadrature(self.dim, 6)        # exact to degree 11 >= 3 + 2 + 3
        phi, dphi = self.tabulate(lam)
        S = np.einsum("p,pai,pbj->abij", w, dphi, dphi)            # avg d_i phi_a d_j phi_b
        bI = np.einsum("p,pai->ai", w, dphi)                       # avg d_i phi_a
        T1 = np.einsum("p,pk,pbi,pa->kiba", w, phi, dphi, phi)     # avg phi_k d_i phi_b phi_a
        M = np.einsum("p,pa,pb->ab", w, phi, phi)
        self._tensors = dict(S=np.ascontiguousarray(S), bI=np.ascontiguousarray(bI),
                             T1=np.ascontiguousarray(T1), M=np.ascontiguousarray(M))
        return self._tensors


def velocity_element(dim, k):
    """The scalar element whose vector version is the reference's velocity space (alfi/solver.py:574-586)."""
    return NodalElement(dim, k, bubble=(k < dim and dim == 3))
