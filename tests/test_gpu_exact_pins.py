"""The DEVICE assembly of a Newton step's operator and residual terms (alfi_level_assemble, alfi_level_assemble_mult,
alfi_level_assemble_supg, alfi_level_supg: csrc/kernels_assemble.hip) against the exact rational derivation of
oracle/exact_pins.py -- not against the product's host assembler.  One- and two-cell meshes with rational vertices, nodes
matched by POSITION; 1e-12 of the largest entry.  -m gpu.

Reference lines: alfi/solver.py:565-568 (velocity block of the Newton linearisation), :204-234 + alfi/stabilisation.py:47-97
(SUPG), :320, 325 (PatchPC.update recomputes the operators from the state)."""
from fractions import Fraction

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import exact_pins as X
from tests import exact_cases as C


def _level(ctx, V):
    """A device level on the node graph of V with zero values and no Dirichlet dofs, prepared for the device assembly."""
    from alfi_amd import _hostlib, hip
    from alfi_amd.problem import BSR
    rowptr, colidx = _hostlib.node_graph(V.cell_nodes, V.num_nodes)
    d = V.dim
    A = BSR(V.num_nodes, V.num_nodes, d, rowptr, colidx, np.zeros((len(colidx), d, d)))
    L = hip.Level(ctx, A, np.zeros(0, dtype=np.int32))
    L.set_assembly(V, rowptr, colidx)
    return L, rowptr, colidx


def _dense(V, rowptr, colidx, vals):
    import scipy.sparse as sp
    n = V.num_nodes * V.dim
    return sp.bsr_matrix((vals, colidx, rowptr), shape=(n, n)).toarray()


@pytest.mark.parametrize("ncell", [1, 2])
@pytest.mark.parametrize("dim,name,args", C.ELEMENTS, ids=[e[1] for e in C.ELEMENTS])
def test_device_operator_terms_are_exact(dim, name, args, ncell):
    """nu K + gamma D + adv N(w) formed cell by cell on the device equals the exact integrals; the matrix-free product
    (alfi_level_assemble_mult) equals exact matrix x vector."""
    from alfi_amd import hip
    mesh, V = C.build_space(dim, args, ncell)
    d = V.dim
    nu, gamma, adv = Fraction(3, 70), Fraction(1250, 3), Fraction(3, 4)
    w = C.rational_field(dim, V)
    wf = np.array([[float(x) for x in r] for r in w])
    ctx = hip.Context(0)
    L, rowptr, colidx = _level(ctx, V)
    st = ctx.vec(wf.ravel())
    for a_ in (adv, Fraction(0)):
        exact = C.exact_operator(dim, name, ncell, V, nu, gamma, a_, w)
        L.assemble(float(nu), float(gamma), float(a_), st if a_ else None, False)
        got = _dense(V, rowptr, colidx, L.get_values())
        assert np.abs(got - exact).max() < 1e-12 * np.abs(exact).max(), (name, ncell, float(a_))
        x = np.random.default_rng(2).standard_normal(V.num_nodes * d)
        dx, dy = ctx.vec(x), ctx.vec(V.num_nodes * d)
        L.assemble_mult(float(nu), float(gamma), float(a_), st if a_ else None, dx, dy)
        ref = exact @ x
        assert np.abs(dy.get() - ref).max() < 1e-12 * np.abs(ref).max(), (name, ncell, float(a_))
    L.close()
    ctx.close()


@pytest.mark.parametrize("ncell", [1, 2])
@pytest.mark.parametrize("dim,name,args", C.ELEMENTS, ids=[e[1] for e in C.ELEMENTS])
def test_device_supg_terms_match_the_exact_basis(dim, name, args, ncell):
    """SUPG on the device: (a) about a CONSTANT state the linearisation is weight * beta x an exact rational matrix and the
    residual vanishes; (b) about a general state both equal the terms formed from the exact nodal basis by an independently
    built Gauss-Jacobi rule (beta is not polynomial); (c) the one-pass refresh nu K + gamma D + N(w) + SUPG is their sum."""
    from alfi_amd import hip
    mesh, V = C.build_space(dim, args, ncell)
    d = V.dim
    nu, gamma = Fraction(3, 70), Fraction(1250, 3)
    weight, magic = 0.05, 9.0
    maps = C.exact_node_maps(dim, name, ncell, V)
    ctx = hip.Context(0)
    L, rowptr, colidx = _level(ctx, V)
    L.set_supg(V)
    n = V.num_nodes * d
    zero = np.zeros((len(colidx), d, d))
    # (a) constant state
    cst = [Fraction(3, 2), Fraction(-2, 3), Fraction(5, 7)][:dim]
    per = []
    for cell in C.cells_of(dim, ncell):
        verts = [C.VERTS[dim][v] for v in cell]
        _, M = X.supg_matrix_constant_state(dim, name, verts, nu, cst)
        h2 = float(4 * X.circumradius_squared(verts))
        beta = (4.0 * float(sum(x * x for x in cst)) / h2 + magic * (4.0 * float(nu) / h2) ** 2) ** -0.5
        m = len(M)
        per.append([[[[weight * beta * float(M[a][i][b][j]) for j in range(d)] for b in range(m)] for i in range(d)] for a in range(m)])
    exact_s = C.scatter(dim, maps, per, V.num_nodes)
    st = ctx.vec(np.tile(np.array([float(x) for x in cst]), V.num_nodes))
    L.update_values(zero)
    dF = ctx.vec(n)
    L.supg(float(nu), weight, magic, st, True, dF)
    got = _dense(V, rowptr, colidx, L.get_values())
    assert np.abs(got - exact_s).max() < 1e-12 * np.abs(exact_s).max(), (name, ncell)
    assert np.abs(dF.get()).max() < 1e-13 * np.abs(exact_s).max()
    # (b) general state
    w = C.rational_field(dim, V)
    wf = np.array([[float(x) for x in r] for r in w])
    deg = 3 if name.endswith("+FB") else int(name[1])
    pts, wts = C.gauss_jacobi_rule(dim, deg + 1)
    perA, perF = [], []
    for cell, m in zip(C.cells_of(dim, ncell), maps):
        verts = [C.VERTS[dim][v] for v in cell]
        _, F, A = X.supg_by_quadrature(dim, name, verts, nu, weight, magic, [wf[g] for g in m], pts, wts)
        perA.append(A)
        perF.append(F)
    exact_a, exact_f = C.scatter(dim, maps, perA, V.num_nodes), C.scatter_vec(dim, maps, perF, V.num_nodes)
    st = ctx.vec(wf.ravel())
    L.update_values(zero)
    dF = ctx.vec(n)
    L.supg(float(nu), weight, magic, st, True, dF)
    got = _dense(V, rowptr, colidx, L.get_values())
    assert np.abs(got - exact_a).max() < 1e-11 * np.abs(exact_a).max(), (name, ncell)
    assert np.abs(dF.get() - exact_f).max() < 1e-11 * np.abs(exact_f).max(), (name, ncell)
    # (c) the one-pass refresh of a stabilised run
    adv = Fraction(1)
    exact_op = C.exact_operator(dim, name, ncell, V, nu, gamma, adv, w) + exact_a
    L.assemble_supg(float(nu), float(gamma), float(adv), st, weight, magic, False)
    got = _dense(V, rowptr, colidx, L.get_values())
    assert np.abs(got - exact_op).max() < 1e-12 * np.abs(exact_op).max(), (name, ncell)
    L.close()
    ctx.close()
