"""The product's finite-element ingredients (and the NumPy oracle's quadrature assembly) against oracle/exact_pins.py, an
exact rational-arithmetic derivation that imports nothing from alfi_amd.  Nodes are matched by POSITION, never by index, so
a wrong node ordering, basis function, quadrature rule or interpolation stencil cannot pass.  CPU only."""
from fractions import Fraction

import numpy as np
import pytest

from oracle import exact_pins as X
from alfi_amd.elements import NodalElement, simplex_quadrature
from alfi_amd.mesh import SimplexMesh, refine
from alfi_amd.fespace import VectorFunctionSpace, nodal_prolongation
from alfi_amd import _hostlib

ELEMENTS = [(2, "P2", (2, 2, False)), (3, "P1+FB", (3, 1, True)), (3, "P2+FB", (3, 2, True)), (3, "P3", (3, 3, False))]
# simplices with rational vertex coordinates, deliberately not axis aligned
SIMPLEX = {2: [(Fraction(1, 7), Fraction(2, 5)), (Fraction(9, 4), Fraction(1, 3)), (Fraction(2, 3), Fraction(11, 6))],
           3: [(Fraction(1, 7), Fraction(2, 5), Fraction(1, 9)), (Fraction(9, 4), Fraction(1, 3), Fraction(-1, 5)),
               (Fraction(2, 3), Fraction(11, 6), Fraction(1, 4)), (Fraction(1, 2), Fraction(3, 5), Fraction(13, 8))]}


def _match(product_bary, exact_nodes):
    """perm[a] = index into exact_nodes of the product's local node a (by barycentric position)."""
    E = np.array([[float(x) for x in n] for n in exact_nodes])
    perm = []
    for p in np.asarray(product_bary):
        d = np.abs(E - p[None, :]).max(axis=1)
        assert d.min() < 1e-12, "product node %r is not a node of the exact element" % (p,)
        perm.append(int(d.argmin()))
    assert sorted(perm) == list(range(len(exact_nodes))), "node sets differ"
    return perm


@pytest.mark.parametrize("dim,name,args", ELEMENTS)
def test_nodal_basis_and_gradients_match_the_exact_dual_basis(dim, name, args):
    el = NodalElement(*args)
    nodes, basis = X.cached_basis(dim, name)
    assert el.nloc == len(nodes)
    perm = _match(el.node_bary, nodes)
    rng = np.random.default_rng(1)
    lam = rng.dirichlet(np.ones(dim + 1), size=7)
    phi, dphi = el.tabulate(lam)
    grads, _ = X.barycentric_gradients(SIMPLEX[dim])
    G = np.array([[float(x) for x in g] for g in grads])                       # (dim+1, dim)
    for a in range(el.nloc):
        b = basis[perm[a]]
        exact = np.array([float(X.p_eval(b, [Fraction(x).limit_denominator(10 ** 12) for x in l])) for l in lam])
        assert np.abs(phi[:, a] - exact).max() < 1e-10
        # physical gradient on the test simplex (the barycentric partials themselves depend on the representation)
        db = [X.p_diff(b, i) for i in range(dim + 1)]
        ge = np.array([[sum(float(X.p_eval(db[i], list(l))) * G[i, x] for i in range(dim + 1)) for x in range(dim)]
                       for l in lam])
        gp = np.einsum("pi,ix->px", dphi[:, a, :], G)
        assert np.abs(gp - ge).max() < 1e-9 * max(1.0, np.abs(ge).max())
    # Kronecker property of the product's basis at its own nodes
    assert np.abs(el.tabulate(el.node_bary)[0] - np.eye(el.nloc)).max() < 1e-13


@pytest.mark.parametrize("dim", [2, 3])
def test_quadrature_integrates_monomials_exactly(dim):
    lam, w = simplex_quadrature(dim, 6)                 # what reference_tensors uses: exact to degree 11
    rng = np.random.default_rng(0)
    for _ in range(20):
        e = tuple(int(k) for k in rng.integers(0, 4, size=dim + 1))
        if sum(e) > 11:
            continue
        exact = float(X.p_average({e: Fraction(1)}))
        assert abs(np.sum(w * np.prod(lam ** np.array(e), axis=1)) - exact) < 1e-14


def _one_cell_space(dim, args):
    coords = np.array([[float(x) for x in v] for v in SIMPLEX[dim]])
    mesh = SimplexMesh(coords, np.arange(dim + 1, dtype=np.int32)[None, :])
    return mesh, VectorFunctionSpace(mesh, NodalElement(*args))


@pytest.mark.parametrize("dim,name,args", ELEMENTS)
def test_element_matrix_of_the_velocity_form_is_exact(dim, name, args):
    """nu (2 sym grad u, grad v) + gamma (cell_avg div u, div v) on one simplex: the product's assembler (reference tensors
    + csrc/host_assemble.cpp) and the NumPy oracle's quadrature assembly against exact integration."""
    from oracle import alfi_oracle as O
    nu, gamma = Fraction(3, 7), Fraction(1250, 3)
    mesh, V = _one_cell_space(dim, args)
    el, d = V.element, V.dim
    nodes, A = X.element_matrix(dim, name, SIMPLEX[dim], nu, gamma)
    perm = _match(el.node_bary, nodes)
    n = el.nloc
    exact = np.zeros((n * d, n * d))
    cn = V.cell_nodes[0]
    for a in range(n):
        for b in range(n):
            for c in range(d):
                for e in range(d):
                    exact[cn[a] * d + c, cn[b] * d + e] = float(A[perm[a]][c][perm[b]][e])
    rowptr, colidx = _hostlib.node_graph(V.cell_nodes, V.num_nodes)
    g, vol = mesh.cell_geometry()
    vals = _hostlib.assemble_bsr(V.cell_nodes, g, vol, el.reference_tensors(), d, rowptr, colidx, nu=float(nu),
                                 gamma=float(gamma))
    import scipy.sparse as sp
    prod = sp.bsr_matrix((vals, colidx, rowptr), shape=(n * d, n * d)).toarray()
    scale = np.abs(exact).max()
    assert np.abs(prod - exact).max() < 1e-12 * scale
    orc = O.assemble_form(V, nu=float(nu), gamma=float(gamma)).toarray()
    assert np.abs(orc - exact).max() < 1e-12 * scale
    assert np.abs(exact - exact.T).max() < 1e-14 * scale            # the form is symmetric


@pytest.mark.parametrize("dim,name,args", ELEMENTS[:3])
def test_nodal_prolongation_rows_are_the_coarse_basis_at_the_fine_nodes(dim, name, args):
    """firedrake.prolong [3P] (alfi/transfer.py:284-286): every row of P holds the coarse nodal basis evaluated at the fine
    node.  One coarse simplex, refined once."""
    mesh, Vc = _one_cell_space(dim, args)
    fine = refine(mesh)
    Vf = VectorFunctionSpace(fine, NodalElement(*args))
    P = nodal_prolongation(Vc, Vf).toarray()
    nodes, basis = X.cached_basis(dim, name)
    perm = _match(Vc.element.node_bary, nodes)
    # barycentric coordinates of every fine node in the coarse simplex
    M = np.concatenate([np.ones((dim + 1, 1)), mesh.coords[mesh.cells[0]]], axis=1)       # rows (1, x_i)
    lam = np.concatenate([np.ones((Vf.num_nodes, 1)), Vf.node_coords], axis=1) @ np.linalg.inv(M)
    assert np.abs(lam.sum(axis=1) - 1).max() < 1e-12 and lam.min() > -1e-12
    values = set()
    for f in range(Vf.num_nodes):
        lf = [Fraction(x).limit_denominator(1000) for x in lam[f]]                         # fine nodes sit on a lattice
        for a in range(Vc.element.nloc):
            ex = X.p_eval(basis[perm[a]], lf)
            assert abs(P[f, Vc.cell_nodes[0][a]] - float(ex)) < 1e-13
            values.add(ex)
    if name == "P2":
        assert values == {Fraction(1), Fraction(3, 8), Fraction(3, 4), Fraction(-1, 8), Fraction(1, 2), Fraction(1, 4),
                          Fraction(0)}
        assert (np.abs(P) > 1e-14).sum(axis=1).max() <= 6


@pytest.mark.parametrize("dim,name,pname,args", [(2, "P2", "P1", (2, 2, False)), (3, "P3", "P2", (3, 3, False))])
def test_scott_vogelius_matrices_are_exact(dim, name, pname, args):
    """The Scott-Vogelius ingredients of config 5 on one simplex against exact integration: the velocity block with the FULL
    grad-div term (solver.py:616; the product's assembler with gamma_full), the discrete divergence against the discontinuous
    P_{k-1} basis and the DG mass matrix (sv.build_sv_pressure_coupling; solver.py:619, 15-38).  And, in rationals, the
    identity behind the augmented Lagrangian of the pair: gamma B^T M^-1 B IS gamma (div u, div v)."""
    import scipy.sparse as sp
    import sympy
    from alfi_amd.sv import build_sv_pressure_coupling
    nu, gamma = Fraction(3, 7), Fraction(1250, 3)
    mesh, V = _one_cell_space(dim, args)
    el, d, n = V.element, V.dim, V.element.nloc
    nodes, pnodes, A, B, M = X.sv_matrices(dim, name, pname, SIMPLEX[dim], nu, gamma)
    perm = _match(el.node_bary, nodes)
    cn = V.cell_nodes[0]
    exact = np.zeros((n * d, n * d))
    for a in range(n):
        for b in range(n):
            for c in range(d):
                for e in range(d):
                    exact[cn[a] * d + c, cn[b] * d + e] = float(A[perm[a]][c][perm[b]][e])
    rowptr, colidx = _hostlib.node_graph(V.cell_nodes, V.num_nodes)
    g, vol = mesh.cell_geometry()
    vals = _hostlib.assemble_bsr(V.cell_nodes, g, vol, el.reference_tensors(), d, rowptr, colidx, nu=float(nu), gamma=0.0,
                                 gamma_full=float(gamma))
    prod = sp.bsr_matrix((vals, colidx, rowptr), shape=(n * d, n * d)).toarray()
    assert np.abs(prod - exact).max() < 1e-12 * np.abs(exact).max()
    # pressure side: match the product's pressure nodes by position as well
    class _L(object):
        pass
    L = _L()
    L.V = V
    Bp, Mp, Mip = build_sv_pressure_coupling(L, zero_bc_columns=False)
    pel = NodalElement(dim, args[1] - 1, False)
    pperm = _match(pel.node_bary, pnodes)
    m = len(pnodes)
    Bex = np.zeros((m, n * d))
    for j in range(m):
        for a in range(n):
            for x in range(d):
                Bex[j, cn[a] * d + x] = float(B[pperm[j]][perm[a]][x])
    Mex = np.array([[float(M[pperm[j]][pperm[l]]) for l in range(m)] for j in range(m)])
    assert np.abs(Bp.toarray() - Bex).max() < 1e-12 * np.abs(Bex).max()
    assert np.abs(Mp.toarray() - Mex).max() < 1e-13 * np.abs(Mex).max()
    assert np.abs(Mip.toarray() @ Mex - np.eye(m)).max() < 1e-10
    # gamma B^T M^-1 B == the grad-div part of A, exactly
    Bm = sympy.Matrix(m, n * dim, lambda j, q: sympy.Rational(B[j][q // dim][q % dim].numerator, B[j][q // dim][q % dim].denominator))
    Mm = sympy.Matrix(m, m, lambda j, l: sympy.Rational(M[j][l].numerator, M[j][l].denominator))
    G = Bm.T * Mm.inv() * Bm * sympy.Rational(gamma.numerator, gamma.denominator)
    _, _, A0, _, _ = X.sv_matrices(dim, name, pname, SIMPLEX[dim], nu, 0)
    for a in range(0, n, 3):
        for b in range(n):
            for c in range(dim):
                for e in range(dim):
                    diff = A[a][c][b][e] - A0[a][c][b][e]
                    assert G[a * dim + c, b * dim + e] == sympy.Rational(diff.numerator, diff.denominator)


@pytest.mark.parametrize("ncell", [1, 2])
@pytest.mark.parametrize("dim,name,args", ELEMENTS[:3])
def test_state_dependent_terms_of_the_host_assembler_are_exact(dim, name, args, ncell):
    """The terms a Newton step re-forms (solver.py:565-568, 204-234; stabilisation.py:47-97) on a one- and a two-cell mesh with
    rational vertices, the product's HOST assembler (csrc/host_assemble.cpp) against oracle/exact_pins.py: nu K + gamma D +
    adv N(w) exactly; the SUPG linearisation about a constant state exactly up to weight * beta (and its vanishing residual);
    the SUPG terms about a general state from the exact basis by an independently built Gauss-Jacobi rule.  Nodes matched by
    position.  The DEVICE assembly meets the same values in tests/test_gpu_exact_pins.py."""
    import scipy.sparse as sp
    from tests import exact_cases as C
    mesh, V = C.build_space(dim, args, ncell)
    d = V.dim
    nu, gamma, adv = Fraction(3, 70), Fraction(1250, 3), Fraction(3, 4)
    w = C.rational_field(dim, V)
    exact = C.exact_operator(dim, name, ncell, V, nu, gamma, adv, w)
    rowptr, colidx = _hostlib.node_graph(V.cell_nodes, V.num_nodes)
    g, vol = mesh.cell_geometry()
    wf = np.array([[float(x) for x in r] for r in w])
    vals = _hostlib.assemble_bsr(V.cell_nodes, g, vol, V.element.reference_tensors(), d, rowptr, colidx, nu=float(nu),
                                 gamma=float(gamma), adv=float(adv), wind=wf)
    prod = sp.bsr_matrix((vals, colidx, rowptr), shape=(V.num_nodes * d,) * 2).toarray()
    assert np.abs(prod - exact).max() < 1e-12 * np.abs(exact).max()
    # cell size: Firedrake's CellSize = 2 x circumradius, exactly
    h = _hostlib.cell_size(mesh)
    for c, cell in enumerate(C.cells_of(dim, ncell)):
        assert abs(h[c] ** 2 - float(4 * X.circumradius_squared([C.VERTS[dim][v] for v in cell]))) < 1e-12 * h[c] ** 2
    # SUPG about a constant state
    maps = C.exact_node_maps(dim, name, ncell, V)
    weight, magic = 0.05, 9.0
    cst = [Fraction(3, 2), Fraction(-2, 3), Fraction(5, 7)][:dim]
    per = []
    for cell in C.cells_of(dim, ncell):
        verts = [C.VERTS[dim][v] for v in cell]
        _, M = X.supg_matrix_constant_state(dim, name, verts, nu, cst)
        h2 = float(4 * X.circumradius_squared(verts))
        beta = (4.0 * float(sum(x * x for x in cst)) / h2 + magic * (4.0 * float(nu) / h2) ** 2) ** -0.5
        n = len(M)
        per.append([[[[weight * beta * float(M[a][i][b][j]) for j in range(d)] for b in range(n)] for i in range(d)] for a in range(n)])
    exact_s = C.scatter(dim, maps, per, V.num_nodes)
    U = np.tile(np.array([float(x) for x in cst]), (V.num_nodes, 1))
    vals = np.zeros((len(colidx), d, d))
    Fh = np.zeros(V.num_nodes * d)
    _hostlib.supg(V, U, float(nu), weight, magic, rowptr, colidx, vals, Fh)
    prod = sp.bsr_matrix((vals, colidx, rowptr), shape=(V.num_nodes * d,) * 2).toarray()
    assert np.abs(prod - exact_s).max() < 1e-12 * np.abs(exact_s).max()
    assert np.abs(Fh).max() < 1e-13 * np.abs(exact_s).max()
    # SUPG about a general state, same rule built independently
    deg = 3 if name.endswith("+FB") else int(name[1])
    pts, wts = C.gauss_jacobi_rule(dim, deg + 1)
    perA, perF = [], []
    for cell, m in zip(C.cells_of(dim, ncell), maps):
        verts = [C.VERTS[dim][v] for v in cell]
        _, F, A = X.supg_by_quadrature(dim, name, verts, nu, weight, magic, [wf[gn] for gn in m], pts, wts)
        perA.append(A)
        perF.append(F)
    exact_a, exact_f = C.scatter(dim, maps, perA, V.num_nodes), C.scatter_vec(dim, maps, perF, V.num_nodes)
    vals = np.zeros((len(colidx), d, d))
    Fh = np.zeros(V.num_nodes * d)
    _hostlib.supg(V, wf, float(nu), weight, magic, rowptr, colidx, vals, Fh)
    prod = sp.bsr_matrix((vals, colidx, rowptr), shape=(V.num_nodes * d,) * 2).toarray()
    assert np.abs(prod - exact_a).max() < 1e-11 * np.abs(exact_a).max()
    assert np.abs(Fh - exact_f).max() < 1e-11 * np.abs(exact_f).max()
