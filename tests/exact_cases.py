"""One- and two-cell meshes with rational vertex coordinates and the EXACT global matrices / vectors of the state-dependent
terms on them (oracle/exact_pins.py), scattered by node POSITION.  Shared by tests/test_exact_pins.py (the product's host
assembler, CPU) and tests/test_gpu_exact_pins.py (the device assembly, -m gpu).  Test infrastructure."""
from fractions import Fraction

import numpy as np

from oracle import exact_pins as X

ELEMENTS = [(2, "P2", (2, 2, False)), (3, "P1+FB", (3, 1, True)), (3, "P2+FB", (3, 2, True))]
# simplices with rational vertex coordinates, deliberately not axis aligned; the second cell shares the facet opposite vertex 0
VERTS = {2: [(Fraction(1, 7), Fraction(2, 5)), (Fraction(9, 4), Fraction(1, 3)), (Fraction(2, 3), Fraction(11, 6)),
             (Fraction(5, 2), Fraction(9, 5))],
         3: [(Fraction(1, 7), Fraction(2, 5), Fraction(1, 9)), (Fraction(9, 4), Fraction(1, 3), Fraction(-1, 5)),
             (Fraction(2, 3), Fraction(11, 6), Fraction(1, 4)), (Fraction(1, 2), Fraction(3, 5), Fraction(13, 8)),
             (Fraction(12, 5), Fraction(7, 4), Fraction(3, 2))]}


def cells_of(dim, ncell):
    """Vertex index lists of the one- or two-cell mesh."""
    cells = [list(range(dim + 1))]
    if ncell == 2:
        cells.append(list(range(1, dim + 2)))
    return cells


def build_space(dim, args, ncell):
    """(mesh, V) of the product on the mesh above (alfi_amd.mesh / fespace)."""
    from alfi_amd.elements import NodalElement
    from alfi_amd.fespace import VectorFunctionSpace
    from alfi_amd.mesh import SimplexMesh
    nv = dim + 1 + (ncell - 1)
    coords = np.array([[float(x) for x in v] for v in VERTS[dim][:nv]])
    mesh = SimplexMesh(coords, np.asarray(cells_of(dim, ncell), dtype=np.int32))
    return mesh, VectorFunctionSpace(mesh, NodalElement(*args))


def exact_node_maps(dim, name, ncell, V):
    """Per cell: global node (of the product's space V) of every node of the exact element, matched by physical position."""
    nodes = X.element_nodes(dim, name)
    maps = []
    for cell in cells_of(dim, ncell):
        verts = [VERTS[dim][v] for v in cell]
        pos = np.array([[float(sum(l * verts[i][x] for i, l in enumerate(lam))) for x in range(dim)] for lam in nodes])
        m = []
        for p in pos:
            d = np.abs(V.node_coords - p[None, :]).max(axis=1)
            assert d.min() < 1e-12, "exact node %r is not a node of the product's space" % (p,)
            m.append(int(d.argmin()))
        assert len(set(m)) == len(m)
        maps.append(m)
    assert sorted(set(sum(maps, []))) == list(range(V.num_nodes)), "node sets differ"
    return maps


def rational_field(dim, V):
    """A rational nodal field, one value per GLOBAL node and component (continuous across the shared facet by construction)."""
    return [[Fraction(3 * g - 7 * x + 2, 5 + (g % 4) + x) for x in range(dim)] for g in range(V.num_nodes)]


def scatter(dim, maps, per_cell, nnode):
    """Dense (nnode dim) x (nnode dim) float matrix from per-cell exact arrays A[a][c][b][d], added by global node."""
    out = np.zeros((nnode * dim, nnode * dim))
    for m, A in zip(maps, per_cell):
        n = len(m)
        for a in range(n):
            for b in range(n):
                for c in range(dim):
                    for e in range(dim):
                        out[m[a] * dim + c, m[b] * dim + e] += float(A[a][c][b][e])
    return out


def scatter_vec(dim, maps, per_cell, nnode):
    out = np.zeros(nnode * dim)
    for m, F in zip(maps, per_cell):
        for a in range(len(m)):
            for c in range(dim):
                out[m[a] * dim + c] += float(F[a][c])
    return out


def exact_operator(dim, name, ncell, V, nu, gamma, adv, w):
    """nu K + gamma D + adv N(w) of the mesh, exactly (then rounded to float): dense matrix in V's numbering."""
    maps = exact_node_maps(dim, name, ncell, V)
    per = []
    for cell, m in zip(cells_of(dim, ncell), maps):
        verts = [VERTS[dim][v] for v in cell]
        _, A = X.element_matrix(dim, name, verts, nu, gamma)
        if adv:
            _, N = X.advection_matrix(dim, name, verts, [w[g] for g in m])
            n = len(m)
            A = [[[[A[a][c][b][e] + Fraction(adv) * N[a][c][b][e] for e in range(dim)] for b in range(n)] for c in range(dim)]
                 for a in range(n)]
        per.append(A)
    return scatter(dim, maps, per, V.num_nodes)


def gauss_jacobi_rule(dim, n):
    """Collapsed Gauss-Jacobi rule with n^dim points on the simplex (exact to degree 2 n - 1), barycentric points and weights
    summing to 1 -- built here from scipy's Jacobi nodes, independently of alfi_amd.elements.simplex_quadrature."""
    from scipy.special import roots_jacobi
    rules = []
    for alpha in range(dim - 1, -1, -1):
        x, w = roots_jacobi(n, alpha, 0)
        rules.append((0.5 * (x + 1.0), w / w.sum()))
    pts, wts = [], []
    if dim == 2:
        for a, wa in zip(*rules[0]):
            for b, wb in zip(*rules[1]):
                x, y = a, b * (1 - a)
                pts.append((1 - x - y, x, y))
                wts.append(wa * wb)
    else:
        for a, wa in zip(*rules[0]):
            for b, wb in zip(*rules[1]):
                for c, wc in zip(*rules[2]):
                    x, y, z = a, b * (1 - a), c * (1 - a) * (1 - b)
                    pts.append((1 - x - y - z, x, y, z))
                    wts.append(wa * wb * wc)
    return pts, wts
