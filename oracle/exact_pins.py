"""Exact (rational-arithmetic) derivation of the finite-element ingredients of the hot path, independent of alfi_amd.

TEST INFRASTRUCTURE.  Nothing here imports the product (alfi_amd/) or the NumPy oracle: node sets, nodal bases, one
element matrix and the nodal prolongation rows are re-derived from their definitions with SymPy / ``fractions`` so that a
wrong basis function, node ordering, quadrature rule or transfer stencil in the product cannot pass both sides.

What is derived, and from which lines of the reference (/root/reference/alfi):

* velocity elements: ``VectorElement(NodalEnrichedElement(P_k, FacetBubble))`` for k < tdim, ``VectorElement(P_k)``
  otherwise (solver.py:574-586), ``[P_k]^d`` for Scott-Vogelius (solver.py:625-630).  All dofs are point evaluations:
  vertices, edge points, face barycentres (bubble.py:43-44, 64-80 hard-codes the change of basis this implies).  The nodal
  basis is the dual basis of those point evaluations on span{P_k} + span{27 l_j l_k l_l}: solved exactly below.
* the velocity-block form ``nu (2 sym grad u, grad v) + gamma (cell_avg(div u), div v)`` (solver.py:565-568,
  transfer.py:319-324): one element matrix on a simplex with rational vertex coordinates, integrated exactly with
  int_K l^alpha = |K| d! alpha! / (|alpha| + d)!.
* the Scott-Vogelius pair (solver.py:612-630): the velocity form with the FULL grad-div term, the discrete divergence against
  the discontinuous P_{k-1} pressure basis and the pressure mass matrix of DGMassInv (solver.py:15-38), exactly on one simplex
  (``sv_matrices``), with the identity gamma B^T M^-1 B = gamma (div u, div v) checked in rationals by the consumer.
* nodal interpolation (firedrake.prolong [3P], transfer.py:284-286): row of a fine node = coarse basis at that node.  For
  2-D P2 the values are {1, 3/8, 3/4, -1/8, 1/2, 1/4, 0} with at most 6 non-zeros per row.

* the state-dependent terms the operator refresh of a Newton step forms (solver.py:565-568, 204-234; stabilisation.py:47-97):
  the linearised advection term about a field of the element space, exactly; the SUPG linearisation about a constant state,
  exactly up to the scalar weight * beta; the SUPG terms about a general state by a quadrature rule handed in (beta is not
  polynomial), from the exact basis.  Consumer: tests/test_gpu_exact_pins.py -- the DEVICE assembly against these.

Consumers: tests/test_exact_pins.py (CPU) compares the product's tabulation, assembled element matrices and prolongation
matrices -- and the NumPy oracle's quadrature assembly -- with these, matching nodes by POSITION, never by index.
"""
import itertools
from fractions import Fraction
from math import factorial

import sympy as sp


# -- polynomials in the barycentric coordinates: {exponent tuple: Fraction} -------------------------------------------------
def p_add(a, b, sb=1):
    out = dict(a)
    for e, c in b.items():
        v = out.get(e, 0) + sb * c
        if v == 0:
            out.pop(e, None)
        else:
            out[e] = v
    return out


def p_scale(a, s):
    return {e: c * s for e, c in a.items()} if s != 0 else {}


def p_mul(a, b):
    out = {}
    for ea, ca in a.items():
        for eb, cb in b.items():
            e = tuple(x + y for x, y in zip(ea, eb))
            v = out.get(e, 0) + ca * cb
            if v == 0:
                out.pop(e, None)
            else:
                out[e] = v
    return out


def p_diff(a, i):
    out = {}
    for e, c in a.items():
        if e[i] > 0:
            f = list(e)
            f[i] -= 1
            out[tuple(f)] = out.get(tuple(f), 0) + c * e[i]
    return out


def p_eval(a, lam):
    s = 0
    for e, c in a.items():
        t = c
        for x, k in zip(lam, e):
            if k:
                t = t * x ** k
        s = s + t
    return s


def p_average(a):
    """Average over the simplex: int_K l^alpha / |K| = d! alpha! / (|alpha| + d)!  (d = number of coordinates - 1)."""
    s = Fraction(0)
    for e, c in a.items():
        d = len(e) - 1
        num = factorial(d)
        for k in e:
            num *= factorial(k)
        s += c * Fraction(num, factorial(sum(e) + d))
    return s


# -- elements ------------------------------------------------------------------------------------------------------------------
def element_nodes(dim, name):
    """Barycentric coordinates (tuples of Fractions) of the point-evaluation dofs.  Order: vertices, edge points (edges as
    sorted vertex pairs, points from the lower to the higher vertex), face barycentres -- consumers must match by position."""
    nv = dim + 1
    unit = lambda i: tuple(Fraction(int(j == i)) for j in range(nv))
    nodes = [unit(i) for i in range(nv)]
    degree = {"P1": 1, "P1+FB": 1, "P2": 2, "P2+FB": 2, "P3": 3}[name]
    for a, b in itertools.combinations(range(nv), 2):
        for s in range(1, degree):
            t = Fraction(s, degree)
            nodes.append(tuple((1 - t) * x + t * y for x, y in zip(unit(a), unit(b))))
    if name.endswith("+FB") or (name == "P3" and dim == 3):
        assert dim == 3
        for face in itertools.combinations(range(nv), 3):
            nodes.append(tuple(Fraction(1, 3) if j in face else Fraction(0) for j in range(nv)))
    if name == "P3" and dim == 2:
        nodes.append(tuple(Fraction(1, 3) for _ in range(nv)))
    return nodes


def element_span(dim, name):
    """A spanning set of the scalar space: homogeneous monomials of degree k in the barycentric coordinates (they span P_k on
    the simplex because sum(l) = 1) plus, for '+FB', the facet bubbles 27 l_j l_k l_l of the four faces."""
    nv = dim + 1
    degree = {"P1": 1, "P1+FB": 1, "P2": 2, "P2+FB": 2, "P3": 3}[name]
    span = []
    for combo in itertools.combinations_with_replacement(range(nv), degree):
        e = [0] * nv
        for i in combo:
            e[i] += 1
        span.append({tuple(e): Fraction(1)})
    if name.endswith("+FB"):
        for face in itertools.combinations(range(nv), 3):
            span.append({tuple(int(j in face) for j in range(nv)): Fraction(27)})
    return span


def nodal_basis(dim, name):
    """(nodes, basis): basis[a] is the polynomial with basis[a](nodes[b]) = delta_ab, obtained by inverting the generalised
    Vandermonde matrix exactly (SymPy rationals)."""
    nodes, span = element_nodes(dim, name), element_span(dim, name)
    assert len(nodes) == len(span), (len(nodes), len(span))
    V = sp.Matrix(len(nodes), len(span), lambda n, s: sp.Rational(*_frac(p_eval(span[s], nodes[n]))))
    C = V.inv()                                   # C[s, a]: coefficient of span_s in basis_a
    basis = []
    for a in range(len(nodes)):
        poly = {}
        for s in range(len(span)):
            c = Fraction(int(C[s, a].p), int(C[s, a].q))
            poly = p_add(poly, p_scale(span[s], c))
        basis.append(poly)
    return nodes, basis


def _frac(x):
    x = Fraction(x)
    return x.numerator, x.denominator


# -- geometry --------------------------------------------------------------------------------------------------------------------
def barycentric_gradients(vertices):
    """vertices: (dim+1) points with rational coordinates.  Returns (grad l_i as lists of Fractions, |K|)."""
    dim = len(vertices) - 1
    M = sp.Matrix(dim + 1, dim + 1, lambda i, j: 1 if j == 0 else sp.Rational(*_frac(vertices[i][j - 1])))
    Minv = M.inv()          # l_i(x) = Minv[0, i] + sum_x Minv[x + 1, i] * x_x
    grads = [[Fraction(int(Minv[x + 1, i].p), int(Minv[x + 1, i].q)) for x in range(dim)] for i in range(dim + 1)]
    det = M.det()
    vol = abs(Fraction(int(det.p), int(det.q))) / factorial(dim)
    return grads, vol


def element_matrix(dim, name, vertices, nu, gamma):
    """Exact element matrix of nu (2 sym grad u, grad v) + gamma (cell_avg(div u), div v) for the vector element [name]^dim
    on the simplex ``vertices``.  Returns (nodes, A) with A[a][c][b][d] (Fractions): row dof = node a, component c."""
    nodes, basis = nodal_basis(dim, name)
    grads, vol = barycentric_gradients(vertices)
    nu, gamma = Fraction(nu), Fraction(gamma)
    n = len(nodes)
    # physical partial derivatives d_x phi_a as polynomials in l
    dphi = [[None] * dim for _ in range(n)]
    for a in range(n):
        dl = [p_diff(basis[a], i) for i in range(dim + 1)]
        for x in range(dim):
            acc = {}
            for i in range(dim + 1):
                acc = p_add(acc, p_scale(dl[i], grads[i][x]))
            dphi[a][x] = acc
    avg = [[p_average(dphi[a][x]) for x in range(dim)] for a in range(n)]
    A = [[[[Fraction(0) for _ in range(dim)] for _ in range(n)] for _ in range(dim)] for _ in range(n)]
    for a in range(n):
        for b in range(n):
            I = [[p_average(p_mul(dphi[a][x], dphi[b][y])) for y in range(dim)] for x in range(dim)]   # avg d_x phi_a d_y phi_b
            gg = sum(I[x][x] for x in range(dim))
            for c in range(dim):
                for d in range(dim):
                    v = nu * vol * ((gg if c == d else 0) + I[d][c]) + gamma * vol * avg[a][c] * avg[b][d]
                    A[a][c][b][d] = v
    return nodes, A


def sv_matrices(dim, name, pname, vertices, nu, gamma):
    """Scott-Vogelius pair [name]^dim - pname^dg on the simplex ``vertices`` (solver.py:612-630), exactly:
    A[a][c][b][d] = nu (2 sym grad u, grad v) + gamma (div u, div v)   (the FULL grad-div term, solver.py:616),
    B[j][a][x] = - int psi_j d_x phi_a   (``- div(u) * q * dx``, solver.py:619),   M[j][l] = int psi_j psi_l   (DGMassInv,
    solver.py:15-38).  Returns (velocity nodes, pressure nodes, A, B, M); div [P_k]^d lies in P_{k-1}, so
    gamma B^T M^-1 B reproduces the grad-div part of A exactly (checked by the consumer in rationals)."""
    nodes, basis = nodal_basis(dim, name)
    pnodes, pbasis = nodal_basis(dim, pname)
    grads, vol = barycentric_gradients(vertices)
    nu, gamma = Fraction(nu), Fraction(gamma)
    n, m = len(nodes), len(pnodes)
    dphi = [[None] * dim for _ in range(n)]
    for a in range(n):
        dl = [p_diff(basis[a], i) for i in range(dim + 1)]
        for x in range(dim):
            acc = {}
            for i in range(dim + 1):
                acc = p_add(acc, p_scale(dl[i], grads[i][x]))
            dphi[a][x] = acc
    A = [[[[Fraction(0) for _ in range(dim)] for _ in range(n)] for _ in range(dim)] for _ in range(n)]
    for a in range(n):
        for b in range(n):
            I = [[p_average(p_mul(dphi[a][x], dphi[b][y])) for y in range(dim)] for x in range(dim)]
            gg = sum(I[x][x] for x in range(dim))
            for c in range(dim):
                for d in range(dim):
                    A[a][c][b][d] = nu * vol * ((gg if c == d else 0) + I[d][c]) + gamma * vol * I[c][d]
    B = [[[-vol * p_average(p_mul(pbasis[j], dphi[a][x])) for x in range(dim)] for a in range(n)] for j in range(m)]
    M = [[vol * p_average(p_mul(pbasis[j], pbasis[l])) for l in range(m)] for j in range(m)]
    return nodes, pnodes, A, B, M


# -- state-dependent terms: advection (exact) and SUPG (exact for a constant state; by a given quadrature rule otherwise) ----------
def _physical_gradients(dim, basis, grads):
    """dphi[a][x]: d phi_a / d x_x as a polynomial in the barycentric coordinates."""
    out = []
    for b in basis:
        dl = [p_diff(b, i) for i in range(dim + 1)]
        row = []
        for x in range(dim):
            acc = {}
            for i in range(dim + 1):
                acc = p_add(acc, p_scale(dl[i], grads[i][x]))
            row.append(acc)
        out.append(row)
    return out


def _to_int(poly):
    """(integer-coefficient polynomial, denominator): products of many polynomials are taken in integers (Fractions are slow)."""
    den = 1
    for c in poly.values():
        c = Fraction(c)
        den = den * c.denominator // _gcd(den, c.denominator)
    return {e: int(Fraction(c) * den) for e, c in poly.items()}, den


def _gcd(a, b):
    while b:
        a, b = b, a % b
    return a


def advection_matrix(dim, name, vertices, w):
    """Exact element matrix of the Newton-linearised advection term of alfi/solver.py:565-568 about the field
    w = sum_k w[k] phi_k (w[k]: ``dim`` rationals per node of element_nodes(dim, name)):
        N[a][c][b][d] = int ((w . grad) phi_b delta_cd + phi_b d_d w_c) phi_a .
    Returns (nodes, N) with Fractions."""
    nodes, basis = cached_basis(dim, name)
    grads, vol = barycentric_gradients(vertices)
    n = len(nodes)
    dphi = _physical_gradients(dim, basis, grads)
    bi = [_to_int(b) for b in basis]
    di = [[_to_int(dphi[a][x]) for x in range(dim)] for a in range(n)]
    # T[k][x][b][a] = int phi_k d_x phi_b phi_a
    T = [[[[None] * n for _ in range(n)] for _ in range(dim)] for _ in range(n)]
    for k in range(n):
        for a in range(k, n):
            pk, dk = bi[k]
            pa, da = bi[a]
            prod = p_mul(pk, pa)
            for b in range(n):
                for x in range(dim):
                    pb, db = di[b][x]
                    v = vol * p_average(p_mul(prod, pb)) / (dk * da * db) if pb else Fraction(0)
                    T[k][x][b][a] = v
                    T[a][x][b][k] = v
    w = [[Fraction(v) for v in wk] for wk in w]
    N = [[[[Fraction(0) for _ in range(dim)] for _ in range(n)] for _ in range(dim)] for _ in range(n)]
    for a in range(n):
        for b in range(n):
            t1 = sum(w[k][x] * T[k][x][b][a] for k in range(n) for x in range(dim))
            for c in range(dim):
                for d in range(dim):
                    # phi_b d_d w_c phi_a = sum_k w[k][c] int phi_b d_d phi_k phi_a = sum_k w[k][c] T[b][d][k][a]
                    N[a][c][b][d] = (t1 if c == d else 0) + sum(w[k][c] * T[b][d][k][a] for k in range(n))
    return nodes, N


def circumradius_squared(vertices):
    """R^2 of the simplex, exactly (the circumcentre solves a linear system with rational coefficients); Firedrake's CellSize
    is 2 R [3P] (alfi/problem.py:46-52 -> stabilisation.py)."""
    dim = len(vertices) - 1
    v0 = vertices[0]
    A = sp.Matrix(dim, dim, lambda i, j: sp.Rational(*_frac(2 * (vertices[i + 1][j] - v0[j]))))
    rhs = sp.Matrix(dim, 1, lambda i, _: sp.Rational(*_frac(sum(Fraction(vertices[i + 1][j]) ** 2 - Fraction(v0[j]) ** 2 for j in range(dim)))))
    c = A.solve(rhs)
    r2 = sum((c[j] - sp.Rational(*_frac(v0[j]))) ** 2 for j in range(dim))
    return Fraction(int(r2.p), int(r2.q))


def supg_matrix_constant_state(dim, name, vertices, nu, c):
    """SUPG linearisation (alfi/stabilisation.py:47-97, solver.py:204-234) about a CONSTANT state u = c, divided by
    weight * beta: with grad u = 0 the strong residual Lu vanishes, beta is a constant and what is left of the Newton
    linearisation of  weight * beta * inner(Lu, dot(grad(v), u))  is
        M[a][i][b][j] = int (dLu)_ij (c . grad phi_a),   (dLu)_ij = - nu (delta_ij Lap phi_b + d_i d_j phi_b) + delta_ij c . grad phi_b
    -- a polynomial integrand: exact.  Returns (nodes, M) with Fractions."""
    nodes, basis = cached_basis(dim, name)
    grads, vol = barycentric_gradients(vertices)
    nu = Fraction(nu)
    c = [Fraction(x) for x in c]
    n = len(nodes)
    dphi = _physical_gradients(dim, basis, grads)
    s = []
    for a in range(n):
        acc = {}
        for x in range(dim):
            acc = p_add(acc, p_scale(dphi[a][x], c[x]))
        s.append(acc)
    # second physical derivatives d_x d_y phi_b
    d2 = []
    for b in range(n):
        rows = []
        for x in range(dim):
            dl = [p_diff(dphi[b][x], i) for i in range(dim + 1)]
            row = []
            for y in range(dim):
                acc = {}
                for i in range(dim + 1):
                    acc = p_add(acc, p_scale(dl[i], grads[i][y]))
                row.append(acc)
            rows.append(row)
        d2.append(rows)
    M = [[[[Fraction(0) for _ in range(dim)] for _ in range(n)] for _ in range(dim)] for _ in range(n)]
    for b in range(n):
        lap = {}
        for x in range(dim):
            lap = p_add(lap, d2[b][x][x])
        for i in range(dim):
            for j in range(dim):
                dL = p_scale(d2[b][i][j], -nu)
                if i == j:
                    dL = p_add(p_add(dL, p_scale(lap, -nu)), s[b])
                for a in range(n):
                    M[a][i][b][j] = vol * p_average(p_mul(dL, s[a]))
    return nodes, M


def supg_by_quadrature(dim, name, vertices, nu, weight, magic, U, lam_pts, wts):
    """The SUPG terms about a general state U (U[k]: ``dim`` floats per node of element_nodes) by a GIVEN quadrature rule
    (barycentric points, weights summing to 1) -- beta is not polynomial --, from the exact nodal basis of this module
    evaluated in floating point: (F[a][i], A[a][i][b][j]) with
        F = weight int beta Lu_i s_a,    Lu = - nu div(2 sym grad u) + (grad u) u,    s_a = u . grad phi_a,
        beta = (4 u.u / h^2 + magic (4 nu / h^2)^2)^(-1/2),   h = 2 x circumradius,
        A = dF / dU[b][j]  (alfi/stabilisation.py:86-97; the derivative of beta included)."""
    import math
    nodes, basis = cached_basis(dim, name)
    grads, vol = barycentric_gradients(vertices)
    n = len(nodes)
    dphi = _physical_gradients(dim, basis, grads)
    d2 = [[[None] * dim for _ in range(dim)] for _ in range(n)]
    for b in range(n):
        for x in range(dim):
            dl = [p_diff(dphi[b][x], i) for i in range(dim + 1)]
            for y in range(dim):
                acc = {}
                for i in range(dim + 1):
                    acc = p_add(acc, p_scale(dl[i], grads[i][y]))
                d2[b][x][y] = acc
    h2 = float(4 * circumradius_squared(vertices))
    nu, vol = float(nu), float(vol)
    F = [[0.0] * dim for _ in range(n)]
    A = [[[[0.0] * dim for _ in range(n)] for _ in range(dim)] for _ in range(n)]
    for lam, wq in zip(lam_pts, wts):
        lam = [float(x) for x in lam]
        ph = [float(p_eval(basis[a], lam)) for a in range(n)]
        gp = [[float(p_eval(dphi[a][x], lam)) for x in range(dim)] for a in range(n)]
        hs = [[[float(p_eval(d2[a][x][y], lam)) for y in range(dim)] for x in range(dim)] for a in range(n)]
        lap = [sum(hs[a][x][x] for x in range(dim)) for a in range(n)]
        u = [sum(ph[a] * U[a][i] for a in range(n)) for i in range(dim)]
        Gu = [[sum(gp[a][x] * U[a][i] for a in range(n)) for x in range(dim)] for i in range(dim)]
        Lu = [sum(-nu * (lap[a] * U[a][i] + sum(hs[a][i][j] * U[a][j] for j in range(dim))) for a in range(n))
              + sum(u[x] * Gu[i][x] for x in range(dim)) for i in range(dim)]
        uu = sum(x * x for x in u)
        beta = 1.0 / math.sqrt(4.0 * uu / h2 + magic * (4.0 * nu / h2) ** 2)
        s = [sum(u[x] * gp[a][x] for x in range(dim)) for a in range(n)]
        wt = float(wq) * vol * weight
        for a in range(n):
            for i in range(dim):
                F[a][i] += wt * beta * Lu[i] * s[a]
                for b in range(n):
                    for j in range(dim):
                        dbeta = -4.0 * beta ** 3 * u[j] * ph[b] / h2
                        dL = -nu * hs[b][i][j] + ph[b] * Gu[i][j] + ((-nu * lap[b] + s[b]) if i == j else 0.0)
                        A[a][i][b][j] += wt * (dbeta * Lu[i] * s[a] + beta * dL * s[a] + beta * Lu[i] * ph[b] * gp[a][j])
    return nodes, F, A


def interpolation_row(dim, name, lam):
    """Values of the nodal basis of ``name`` at the point with barycentric coordinates ``lam`` (floats or Fractions): the
    row of the nodal prolongation matrix of a fine node located there.  Returns (nodes, values)."""
    nodes, basis = nodal_basis(dim, name)
    return nodes, [p_eval(b, lam) for b in basis]


_CACHE = {}


def cached_basis(dim, name):
    key = (dim, name)
    if key not in _CACHE:
        _CACHE[key] = nodal_basis(dim, name)
    return _CACHE[key]
