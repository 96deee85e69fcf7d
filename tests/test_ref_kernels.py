"""Pins against the reference's OWN native code: the five C kernels of alfi/bubble.py, compiled where they lie into
oracle/_ref/libalfi_ref_bubble.so by oracle/build_ref.py (skipped when neither /root/reference nor the prebuilt library
is there).

1. the oracle's split / combine stencils (oracle.alfi_oracle.bubble_prolong_literal) are the reference kernels';
2. BubbleTransfer.prolong (bubble.py:233-265) driven cell by cell through the REFERENCE kernels -- only the par_loop
   orchestration, the facet rescaling (bubble.py:29-39) and the P1 / bubble sub-space prolongations (firedrake.prolong
   [3P]) are restated -- equals the oracle's literal version and the generator's sparse prolongation matrix, i.e. the
   matrix the HIP path multiplies with (alfi_prolong);
3. BubbleTransfer.restrict (bubble.py:204-231) through the reference's adjoint kernels equals the transposed matrix the
   HIP path uses for the robust restriction."""
import ctypes
import os

import numpy as np
import pytest

from alfi_amd.problem import ThreeDimLidDrivenCavityProblem, build_hierarchy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dp = ctypes.POINTER(ctypes.c_double)


@pytest.fixture(scope="module")
def ref():
    from oracle import build_ref
    path = build_ref.build()
    if path is None or not os.path.exists(path):
        pytest.skip("reference kernels not built (no /root/reference and no prebuilt oracle/_ref)")
    lib = ctypes.CDLL(path)
    for name in ("split", "splitadj", "combine", "combineadj", "count"):
        getattr(lib, name).restype = None
        getattr(lib, name).argtypes = [dp, dp, dp]
    return lib


def _p(a):
    return a.ctypes.data_as(dp)


def test_split_combine_stencils_are_the_reference_kernels(ref):
    rng = np.random.default_rng(0)
    a = np.vstack([np.eye(4), np.zeros((4, 4))])                                      # as restated in the oracle
    b = np.vstack([-(np.ones((4, 4)) - np.eye(4)) / 3.0, np.eye(4)])
    ac = np.hstack([np.eye(4), (np.ones((4, 4)) - np.eye(4)) / 3.0])
    bc = np.hstack([np.zeros((4, 4)), np.eye(4)])
    both = rng.standard_normal((8, 3))
    p1, fb = np.zeros((4, 3)), np.zeros((4, 3))
    ref.split(_p(p1), _p(fb), _p(both))
    assert np.allclose(p1, a.T @ both, atol=1e-15) and np.allclose(fb, b.T @ both, atol=1e-15)
    p1, fb = rng.standard_normal((4, 3)), rng.standard_normal((4, 3))
    out = np.zeros((8, 3))
    ref.combine(_p(p1), _p(fb), _p(out))
    assert np.allclose(out, ac.T @ p1 + bc.T @ fb, atol=1e-15)
    # combine after split is the identity on a cell (nodal -> hierarchical -> nodal)
    q1, qb = np.zeros((4, 3)), np.zeros((4, 3))
    ref.split(_p(q1), _p(qb), _p(both))
    back = np.zeros((8, 3))
    ref.combine(_p(q1), _p(qb), _p(back))
    assert np.allclose(back, both, atol=1e-14)
    # the adjoint kernels are the transposes
    out = np.zeros((8, 3))
    ref.splitadj(_p(p1), _p(fb), _p(out))
    assert np.allclose(out, a @ p1 + b @ fb, atol=1e-15)
    q1, qb = np.zeros((4, 3)), np.zeros((4, 3))
    ref.combineadj(_p(both), _p(q1), _p(qb))
    assert np.allclose(q1, ac @ both, atol=1e-15) and np.allclose(qb, bc @ both, atol=1e-15)
    c1, c2, c3 = np.zeros((8, 3)), np.zeros((4, 3)), np.zeros((4, 3))
    ref.count(_p(c1), _p(c2), _p(c3))
    assert (c1 == 1).all() and (c2 == 1).all() and (c3 == 1).all()


def _subspace_prolongations(Vc, Vf):
    from alfi_amd.fespace import nodal_prolongation
    from alfi_amd.elements import NodalElement
    mc, mf = Vc.mesh, Vf.mesh
    p1 = NodalElement(3, 1, False)
    P1 = nodal_prolongation(Vc, Vf, p1, p1, mf.cells, mc.cells, mf.num_vertices, mc.num_vertices)

    class _FB(object):
        node_bary = Vc.element.node_bary[-4:]

        @staticmethod
        def tabulate(lam):
            return NodalElement._bubbles(np.atleast_2d(lam))
    PF = nodal_prolongation(Vc, Vf, _FB, _FB, mf.cell_faces, mc.cell_faces, mf.num_faces, mc.num_faces)
    return P1, PF


def _rescale(mc, fbc):
    """fbc <- ainv * assemble(L): b_F + (1/0.625 - 1)(b_F . n_F) n_F (bubble.py:29-39, 251-253)."""
    from alfi_amd.fespace import _facet_normals
    nrm = _facet_normals(mc)
    return fbc + (1.0 / 0.625 - 1.0) * (fbc * nrm).sum(axis=1)[:, None] * nrm


def _counts(ref, V):
    m = V.mesh
    cv, cf, cp = np.zeros((V.num_nodes, 3)), np.zeros((m.num_faces, 3)), np.zeros((m.num_vertices, 3))
    for c in range(m.num_cells):
        a, b, d = np.zeros((8, 3)), np.zeros((4, 3)), np.zeros((4, 3))
        ref.count(_p(a), _p(b), _p(d))
        cv[V.cell_nodes[c]] += a
        cf[m.cell_faces[c]] += b
        cp[m.cells[c]] += d
    return cv, cf, cp


def prolong_with_reference_kernels(ref, Vc, Vf, coarse):
    mc, mf = Vc.mesh, Vf.mesh
    both = coarse.reshape(-1, 3)
    p1c, fbc = np.zeros((mc.num_vertices, 3)), np.zeros((mc.num_faces, 3))
    for c in range(mc.num_cells):                                                     # par_loop(split_kernel)  :238-241
        a, b = np.zeros((4, 3)), np.zeros((4, 3))
        ref.split(_p(a), _p(b), _p(np.ascontiguousarray(both[Vc.cell_nodes[c]])))
        p1c[mc.cells[c]] += a
        fbc[mc.cell_faces[c]] += b
    _, cfc, cpc = _counts(ref, Vc)
    p1c /= cpc                                                                          # :242-243
    fbc /= cfc
    fbc = _rescale(mc, fbc)                                                             # :250-252
    P1, PF = _subspace_prolongations(Vc, Vf)
    p1f, fbf = P1 @ p1c, PF @ fbc                                                       # :255-256
    fine = np.zeros((Vf.num_nodes, 3))
    for c in range(mf.num_cells):                                                       # par_loop(combine_kernel) :260-263
        out = np.zeros((8, 3))
        ref.combine(_p(np.ascontiguousarray(p1f[mf.cells[c]])), _p(np.ascontiguousarray(fbf[mf.cell_faces[c]])), _p(out))
        fine[Vf.cell_nodes[c]] += out
    cvf, _, _ = _counts(ref, Vf)
    return (fine / cvf).ravel()                                                         # :264


def restrict_with_reference_kernels(ref, Vc, Vf, fine):
    mc, mf = Vc.mesh, Vf.mesh
    cvf, _, _ = _counts(ref, Vf)
    _, cfc, cpc = _counts(ref, Vc)
    f = fine.reshape(-1, 3) / cvf                                                       # :209 (in place in the reference)
    p1f, fbf = np.zeros((mf.num_vertices, 3)), np.zeros((mf.num_faces, 3))
    for c in range(mf.num_cells):                                                       # par_loop(combine_kernel_adj) :210-214
        a, b = np.zeros((4, 3)), np.zeros((4, 3))
        ref.combineadj(_p(np.ascontiguousarray(f[Vf.cell_nodes[c]])), _p(a), _p(b))
        p1f[mf.cells[c]] += a
        fbf[mf.cell_faces[c]] += b
    P1, PF = _subspace_prolongations(Vc, Vf)
    p1c, fbc = P1.T @ p1f, PF.T @ fbf                                                   # firedrake.restrict       :216-217
    fbc = _rescale(mc, fbc)                                                             # :219-221
    p1c /= cpc                                                                          # :225-226
    fbc /= cfc
    coarse = np.zeros((Vc.num_nodes, 3))
    for c in range(mc.num_cells):                                                       # par_loop(split_kernel_adj) :227-230
        out = np.zeros((8, 3))
        ref.splitadj(_p(np.ascontiguousarray(p1c[mc.cells[c]])), _p(np.ascontiguousarray(fbc[mc.cell_faces[c]])), _p(out))
        coarse[Vc.cell_nodes[c]] += out
    return coarse.ravel()


def test_bubble_transfer_through_reference_kernels(ref):
    from oracle import alfi_oracle as O
    lv, tr = build_hierarchy(ThreeDimLidDrivenCavityProblem(2), 1, 1, Re=10.0)
    Vc, Vf, T = lv[0].V, lv[1].V, tr[0]
    rng = np.random.default_rng(1)
    uc = rng.standard_normal(Vc.num_dofs)
    got = prolong_with_reference_kernels(ref, Vc, Vf, uc)
    lit = O.bubble_prolong_literal(Vc, Vf, uc)
    mat = T.P.to_scipy() @ uc
    assert np.abs(got - lit).max() < 1e-13 * np.abs(got).max()
    assert np.abs(got - mat).max() < 1e-13 * np.abs(got).max()
    rf = rng.standard_normal(Vf.num_dofs)
    gotr = restrict_with_reference_kernels(ref, Vc, Vf, rf)
    matr = T.PT.to_scipy() @ rf
    assert np.abs(gotr - matr).max() < 1e-13 * np.abs(gotr).max()


@pytest.mark.gpu
def test_hip_prolong_and_restrict_against_reference_kernels(ref):
    """The device path directly against the reference's own C kernels: alfi_prolong / alfi_restrict with gamma = 0 (then
    the Schoeberl correction vanishes, transfer.py:259 / :274, and what is left is BubbleTransfer.prolong / .restrict,
    bubble.py:204-265) on ldc3d [P1+FB]^3, compared with the cell-by-cell evaluation through the reference kernels."""
    from alfi_amd import hip
    lv, tr = build_hierarchy(ThreeDimLidDrivenCavityProblem(2), 1, 1, Re=10.0)
    Vc, Vf, T = lv[0].V, lv[1].V, tr[0]
    ctx = hip.Context(0)
    dc, df = hip.Level(ctx, lv[0].A, lv[0].bc_dofs), hip.Level(ctx, lv[1].A, lv[1].bc_dofs)
    dt = hip.Transfer(ctx, dc, df, T)
    dt.update(T.nu, 0.0)
    rng = np.random.default_rng(2)
    uc = rng.standard_normal(Vc.num_dofs)
    want = prolong_with_reference_kernels(ref, Vc, Vf, uc)
    want[lv[1].bc_dofs] = 0.0                                  # alfi_prolong zeroes the fine Dirichlet dofs
    xc, xf = ctx.vec(uc), ctx.vec(Vf.num_dofs)
    dt.prolong(xc, xf)
    got = xf.get()
    assert np.abs(got - want).max() < 1e-13 * np.abs(want).max()
    rf = rng.standard_normal(Vf.num_dofs)
    wantr = restrict_with_reference_kernels(ref, Vc, Vf, rf.copy())
    wantr[lv[0].bc_dofs] = 0.0
    yf, yc = ctx.vec(rf), ctx.vec(Vc.num_dofs)
    dt.restrict(yf, yc, robust=True)
    gotr = yc.get()
    assert np.abs(gotr - wantr).max() < 1e-13 * np.abs(wantr).max()
    dt.close()
    dc.close()
    df.close()
    ctx.close()
