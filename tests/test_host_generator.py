"""CPU tests of the host-side generator (mesh, spaces, patches, operators, transfers) against the oracle's independent
restatements, plus the mathematical properties SURVEY.md section 4 lists as the pins for this (test-less) reference."""
import numpy as np
import pytest
import scipy.sparse as sp

from alfi_amd.problem import (TwoDimLidDrivenCavityProblem, ThreeDimLidDrivenCavityProblem, build_hierarchy)
from alfi_amd.fespace import nodal_prolongation, skeleton_node_mask
from oracle import alfi_oracle as O

CASES = [("2d-P2", lambda: TwoDimLidDrivenCavityProblem(2), 2),
         ("3d-P1FB", lambda: ThreeDimLidDrivenCavityProblem(1), 1),
         ("3d-P2FB", lambda: ThreeDimLidDrivenCavityProblem(1), 2)]


@pytest.fixture(scope="module", params=CASES, ids=[c[0] for c in CASES])
def hier(request):
    name, mk, k = request.param
    prob = mk()
    lv, tr = build_hierarchy(prob, 1, k, Re=50)
    return prob, k, lv, tr


def test_sizes_match_survey():
    """Interior star sizes 14 / 111 / 153 and coarse-cell interior blocks 6 / 24 / 27."""
    for prob, k, npatch, m in [(TwoDimLidDrivenCavityProblem(4), 2, 14, 6),
                               (ThreeDimLidDrivenCavityProblem(2), 1, 111, 24),
                               (ThreeDimLidDrivenCavityProblem(2), 2, 153, 27)]:
        lv, tr = build_hierarchy(prob, 1, k, Re=10)
        assert np.diff(lv[-1].patch_ptr).max() == npatch
        assert tr[-1].blk_dofs.shape[1] == m
        assert tr[-1].blk_dofs.shape[0] == lv[0].V.mesh.num_cells


def test_dof_counts_3d():
    """V=(N+1)^3, E=7N^3+9N^2+3N, F=12N^3+6N^2 for the Kuhn box."""
    from alfi_amd.mesh import box_mesh
    N = 3
    m = box_mesh(N, N, N, 2, 2, 2)
    assert m.num_vertices == (N + 1) ** 3
    assert m.num_edges == 7 * N ** 3 + 9 * N ** 2 + 3 * N
    assert m.num_faces == 12 * N ** 3 + 6 * N ** 2
    assert m.num_cells == 6 * N ** 3


def test_refinement_reproduces_structured_mesh():
    """Bey refinement of the Kuhn box with N cubes is the Kuhn box with 2N cubes (same cells as vertex-coordinate sets)."""
    from alfi_amd.mesh import box_mesh, refine, rectangle_mesh
    for coarse, fine in [(box_mesh(2, 2, 2, 2, 2, 2), box_mesh(4, 4, 4, 2, 2, 2)),
                         (rectangle_mesh(3, 3, 2, 2), rectangle_mesh(6, 6, 2, 2))]:
        r = refine(coarse)
        assert r.num_cells == fine.num_cells and r.num_vertices == fine.num_vertices

        def key(m):
            c = np.round(m.coords[m.cells] * 1000).astype(np.int64)            # (nc, d+1, d)
            c = np.array([sorted(map(tuple, cell)) for cell in c]).reshape(m.num_cells, -1)
            return set(map(tuple, c))
        assert key(r) == key(fine)


def test_star_patches_match_literal(hier):
    prob, k, lv, tr = hier
    L = lv[-1]
    V = L.V
    lit = O.star_patches_literal(V)
    assert len(lit) == len(L.patch_ptr) - 1
    order = np.argsort([V.vertex_nodes[v] for v, _ in lit])
    for p, i in enumerate(order):
        v, dofs = lit[i]
        assert L.patch_seeds[p] == v
        assert np.array_equal(dofs, L.patch_dofs[L.patch_ptr[p]:L.patch_ptr[p + 1]])


def test_assembly_matches_quadrature_oracle(hier):
    prob, k, lv, tr = hier
    for L in lv:
        V = L.V
        w = prob.driver(V.node_coords)
        A_o = O.apply_bcs_matrix(O.assemble_form(V, nu=L.nu, gamma=L.gamma, adv=1.0, wind=w), L.bc_dofs)
        A_h = L.A.to_scipy().tocsr()
        assert abs(A_o - A_h).max() <= 1e-12 * abs(A_o).max()


def test_symmetric_parts_spd(hier):
    prob, k, lv, tr = hier
    V = lv[-1].V
    K = O.assemble_form(V, nu=1.0)
    D = O.assemble_form(V, gamma=1.0)
    assert abs(K - K.T).max() < 1e-12 * abs(K).max()
    assert abs(D - D.T).max() < 1e-12 * abs(D).max()
    x = np.random.default_rng(1).standard_normal(V.num_dofs)
    assert x @ (K @ x) > 0 and x @ (D @ x) >= -1e-10
    # rigid translations are in the kernel of both
    assert abs(K @ np.ones(V.num_dofs)).max() < 1e-9 * abs(K).max()
    assert abs(D @ np.ones(V.num_dofs)).max() < 1e-9 * abs(D).max()


def test_nodal_prolongation_reproduces_polynomials(hier):
    prob, k, lv, tr = hier
    Vc, Vf = lv[0].V, lv[1].V
    P = nodal_prolongation(Vc, Vf)

    def poly(x, deg):
        return (1 + x[:, 0] + 2 * x[:, 1]) ** deg + x[:, -1] ** deg
    for deg in range(1, Vf.element.degree + 1):
        assert abs(P @ poly(Vc.node_coords, deg) - poly(Vf.node_coords, deg)).max() < 1e-12


def test_bubble_matrix_matches_literal_kernels():
    prob = ThreeDimLidDrivenCavityProblem(1)
    lv, tr = build_hierarchy(prob, 1, 1, Re=50)
    u = np.random.default_rng(0).standard_normal(lv[0].V.num_dofs)
    ref = O.bubble_prolong_literal(lv[0].V, lv[1].V, u)
    assert abs(tr[-1].P.to_scipy() @ u - ref).max() < 1e-13 * abs(ref).max()


def test_bubble_fix_scales_normal_flux():
    """The rescaling multiplies the normal component of every coarse bubble dof by 1/0.625 = 1.6 and keeps the
    tangential part (bubble.py:4-6, 36): check P_bubble - P_nodal acts only through that."""
    from alfi_amd.fespace import bubble_prolongation, _facet_normals
    prob = ThreeDimLidDrivenCavityProblem(1)
    lv, _ = build_hierarchy(prob, 1, 1, Re=50)
    Vc, Vf = lv[0].V, lv[1].V
    Pb = bubble_prolongation(Vc, Vf)
    Pn = sp.kron(nodal_prolongation(Vc, Vf), sp.identity(3))
    # a coarse function that is a pure tangential bubble on one facet: both transfers agree
    nrm = _facet_normals(Vc.mesh)
    f = 5
    t = np.cross(nrm[f], [0.3, -0.2, 0.9])
    u = np.zeros((Vc.num_nodes, 3))
    u[Vc.face_nodes[f]] = t
    assert abs(Pb @ u.ravel() - Pn @ u.ravel()).max() < 1e-13
    # pure normal bubble: bubble transfer = 1.6 x nodal transfer
    u[Vc.face_nodes[f]] = nrm[f]
    assert abs(Pb @ u.ravel() - 1.6 * (Pn @ u.ravel())).max() < 1e-13


def _oracle_transfer(lv, tr):
    return O.oracle_transfer(tr[-1], lv[-1], schoeberl_restriction=True).st


def test_schoeberl_restrict_is_transpose_of_prolong(hier):
    prob, k, lv, tr = hier
    st = _oracle_transfer(lv, tr)
    rng = np.random.default_rng(2)
    u = rng.standard_normal(lv[0].n)
    r = rng.standard_normal(lv[1].n)
    lhs, rhs = st.prolong(u) @ r, u @ st.restrict(r)
    assert abs(lhs - rhs) < 1e-10 * max(abs(lhs), 1.0)


def test_schoeberl_prolongation_preserves_cell_divergence(hier):
    """Defining property of the robust prolongation: the divergence seen by the fine P0 space of P~ u_H equals the
    coarse cell average of div u_H up to O(nu/gamma), whereas plain interpolation does not achieve that for the
    enriched 3-D elements and leaves O(1) defects for general data."""
    prob, k, lv, tr = hier
    Vc, Vf = lv[0].V, lv[1].V
    st = _oracle_transfer(lv, tr)
    rng = np.random.default_rng(3)
    u = rng.standard_normal(Vc.num_dofs)
    u[lv[0].bc_dofs] = 0.0

    def cell_div(V, x):
        g, vol = V.mesh.cell_geometry()
        bI = V.element.reference_tensors()["bI"]
        b = np.einsum("cix,ai->cax", g, bI)                    # cell average of d_x phi_a
        return np.einsum("cax,cax->c", b, x.reshape(-1, V.dim)[V.cell_nodes])
    dc = cell_div(Vc, u)
    df = cell_div(Vf, st.prolong(u))
    defect = np.abs(df - dc[Vf.mesh.parent_cell]).max()
    assert defect < 50 * (tr[-1].nu / tr[-1].gamma) * max(np.abs(dc).max(), 1.0) + 1e-9


def test_coarse_blocks_are_disjoint_and_off_skeleton(hier):
    prob, k, lv, tr = hier
    V = lv[1].V
    blk = tr[-1].blk_dofs
    assert len(np.unique(blk)) == blk.size
    skel = np.repeat(skeleton_node_mask(V), V.dim)
    assert not skel[blk].any()
    # every non-skeleton dof is in some block; all Dirichlet dofs are on the skeleton
    assert blk.size == (~skel).sum()
    assert skel[lv[1].bc_dofs].all()


def test_bsr_helpers():
    from alfi_amd.problem import BSR
    rng = np.random.default_rng(0)
    M = sp.random(12, 8, density=0.4, random_state=1, format="csr")
    B = BSR.from_scipy(sp.kron(M, np.arange(1, 5).reshape(2, 2)), 2)
    assert abs(B.transpose().to_scipy() - B.to_scipy().T).max() == 0
    rows = np.array([5, 0, 3])
    sub = B.select_rows(rows).to_scipy().toarray()
    full = B.to_scipy().toarray().reshape(12, 2, -1)
    assert np.array_equal(sub, full[rows].reshape(6, -1))
    # a range of block rows as views of the same arrays
    rr = B.row_range(3, 9)
    assert np.array_equal(rr.to_scipy().toarray(), full[3:9].reshape(12, -1))
    assert np.shares_memory(rr.vals, B.vals) and np.shares_memory(rr.colidx, B.colidx)


def test_sparsity_only_hierarchy_and_event_names():
    """build_hierarchy(operator_values=False): the same integers, no operator values (they are formed on the device,
    tests/test_gpu_assemble.py); the profile classes carry the PETSc event names of the reference's report (alfi/driver.py:80)."""
    from alfi_amd import _lib
    full, _ = build_hierarchy(ThreeDimLidDrivenCavityProblem(2), 1, 1, Re=0.0)
    bare, _ = build_hierarchy(ThreeDimLidDrivenCavityProblem(2), 1, 1, Re=0.0, operator_values=False)
    for a, b in zip(full, bare):
        assert b.A.vals is None and a.A.vals is not None
        assert np.array_equal(a.A.rowptr, b.A.rowptr) and np.array_equal(a.A.colidx, b.A.colidx) and a.A.nnzb == b.A.nnzb
    assert set(_lib.PETSC_EVENT_NAMES) == set(_lib.EVENTS)
    for name in ("PCPATCHApply", "PCPATCHScatter", "PCPatchComputeOp", "MatMult", "SchoeberlProlong", "SchoeberlRestrict", "MatSolve"):
        assert name in _lib.PETSC_EVENT_NAMES.values()      # names of driver.py:80


@pytest.mark.parametrize("mk,k", [(lambda: TwoDimLidDrivenCavityProblem(3), 2),
                                  (lambda: ThreeDimLidDrivenCavityProblem(2), 1),
                                  (lambda: ThreeDimLidDrivenCavityProblem(2), 2)])
def test_divergence_matrix_reproduces_the_augmented_lagrangian_term(mk, k):
    """gamma (cell_avg div u, div v) = gamma B^T M_p^-1 B with P0 pressures (solver.py:565-568):
    the level operator at two values of gamma differs by exactly that on the free dofs; B annihilates nothing else."""
    from alfi_amd.problem import build_pressure_coupling
    import scipy.sparse as sp
    lv1, _ = build_hierarchy(mk(), 1, k, Re=10.0, gamma=0.0)
    lv2, _ = build_hierarchy(mk(), 1, k, Re=10.0, gamma=7.0)
    L = lv2[-1]
    B, vol = build_pressure_coupling(L)
    diff = (L.A.to_scipy() - lv1[-1].A.to_scipy()).tocsr()
    ref = (7.0 * (B.T @ sp.diags(1.0 / vol) @ B)).tocsr()
    assert abs(diff - ref).max() < 1e-11 * abs(ref).max()
    assert np.isclose(vol.sum(), 2.0 ** L.V.dim)
    # divergence of a linear field u = (x, 0, ..): -int div u = -|cell| on cells away from the Dirichlet boundary
    u = np.zeros((L.V.num_nodes, L.V.dim))
    u[:, 0] = L.V.node_coords[:, 0]
    Bfull_u = B @ u.ravel()
    interior = ~L.V.bc_node_mask[L.V.cell_nodes].any(axis=1)
    assert interior.any() and np.allclose(Bfull_u[interior], -vol[interior])


def test_gmsh_reader_round_trip_and_channel_problem(tmp_path):
    """gmsh 2.2 ASCII meshes (the format of the reference's examples/bfs3d/coarse*.msh): write the structured channel with
    the reference's physical tags (1 inflow x = 0, 2 outflow x = 10, 3 walls; bfs3d.py:23-26), read it back -- same
    geometry, same boundary -- and run the problem / hierarchy set-up on it, with shuffled node ids and cell orientations."""
    from alfi_amd.mesh import bfs3d_mesh, read_gmsh, write_gmsh
    from alfi_amd.problem import ThreeDimBackwardsFacingStepProblem
    from alfi_amd.fespace import VectorFunctionSpace
    from alfi_amd.elements import NodalElement
    m = bfs3d_mesh(1)
    path = str(tmp_path / "channel.msh")

    def tags(c):
        return np.where(c[:, 0] < 1e-12, 1, np.where(c[:, 0] > 10 - 1e-12, 2, 3))
    write_gmsh(m, path, tags)
    r = read_gmsh(path)
    assert r.num_cells == m.num_cells and r.num_vertices == m.num_vertices
    assert np.isclose(r.cell_geometry()[1].sum(), 19.0)                           # 10 x 2 x 1 minus the 1 x 1 x 1 step
    assert len(r.boundary_tags) == len(r.boundary_facets)
    cent = {k: r.coords[list(k)].mean(axis=0) for k in r.boundary_tags}
    assert all((t == 2) == (abs(cent[k][0] - 10.0) < 1e-12) for k, t in r.boundary_tags.items())
    # a file with permuted node ids, non-contiguous numbering and flipped tets reads to the same mesh geometry
    lines = open(path).read().split("\\n")
    rng = np.random.default_rng(0)
    nn = m.num_vertices
    new_id = rng.permutation(nn) * 3 + 7
    out, section = [], None
    for l in lines:
        if l.startswith("$"):
            section = l
            out.append(l)
            continue
        t = l.split()
        if section == "$Nodes" and len(t) == 4:
            out.append("%d %s %s %s" % (new_id[int(t[0]) - 1], t[1], t[2], t[3]))
        elif section == "$Elements" and len(t) > 4:
            nt = int(t[2])
            vs = [str(new_id[int(v) - 1]) for v in t[3 + nt:]]
            if t[1] == "4" and int(t[0]) % 2 == 0:
                vs[0], vs[1] = vs[1], vs[0]
            out.append(" ".join(t[:3 + nt] + vs))
        else:
            out.append(l)
    path2 = str(tmp_path / "shuffled.msh")
    open(path2, "w").write("\\n".join(out))
    r2 = read_gmsh(path2)
    assert r2.num_cells == m.num_cells and np.isclose(r2.cell_geometry()[1].sum(), 19.0)
    prob = ThreeDimBackwardsFacingStepProblem(msh=path2)
    mh = prob.mesh_hierarchy("uniform", 1)
    assert mh[1].num_cells == 8 * m.num_cells and np.isclose(mh[1].cell_geometry()[1].sum(), 19.0)
    V = VectorFunctionSpace(mh[0], NodalElement(3, 2, False), dirichlet=prob.dirichlet_facets)
    Vs = VectorFunctionSpace(m, NodalElement(3, 2, False), dirichlet=prob.dirichlet_facets)
    assert V.num_nodes == Vs.num_nodes and len(V.bc_nodes) == len(Vs.bc_nodes)
    # the Dirichlet boundary is chosen geometrically (x < 10), the reference chooses by tag (bfs3d.py:23-26): a
    # mesh whose tags say otherwise -- here: inflow and outflow swapped -- is refused instead of silently mis-conditioned
    path3 = str(tmp_path / "swapped.msh")
    write_gmsh(m, path3, lambda c: np.where(c[:, 0] < 1e-12, 2, np.where(c[:, 0] > 10 - 1e-12, 1, 3)))
    with pytest.raises(ValueError, match="disagrees with the geometric choice"):
        ThreeDimBackwardsFacingStepProblem(msh=path3).mesh()


@pytest.mark.parametrize("dim,k", [(2, 2), (3, 1), (3, 2)])
def test_contributor_lists_of_the_device_assembly(dim, k):
    """_hostlib.contributors (the gather lists of alfi_level_set_assembly): every (cell, a, b) triple appears exactly once, in
    the block (node a, node b) of the level's sparsity, cells ascending inside a block; summing host element matrices through
    the lists reproduces the assembled operator."""
    from alfi_amd import _hostlib
    from alfi_amd.problem import TwoDimLidDrivenCavityProblem, ThreeDimLidDrivenCavityProblem, build_hierarchy
    prob = TwoDimLidDrivenCavityProblem(3) if dim == 2 else ThreeDimLidDrivenCavityProblem(2)
    lv, _ = build_hierarchy(prob, 1, k, Re=10.0)
    L = lv[-1]
    V, A = L.V, L.A
    cn = V.cell_nodes
    nloc = cn.shape[1]
    cptr, ccell, cba = _hostlib.contributors(cn, V.num_nodes, A.rowptr, A.colidx)
    assert cptr[0] == 0 and cptr[-1] == cn.shape[0] * nloc * nloc and (np.diff(cptr) >= 1).all()
    rows = np.repeat(np.arange(A.nbrows), np.diff(A.rowptr))
    blk = np.repeat(np.arange(A.nnzb), np.diff(cptr))
    a, b = cba.astype(np.int64) % nloc, cba.astype(np.int64) // nloc
    assert np.array_equal(cn[ccell, a], rows[blk]) and np.array_equal(cn[ccell, b], A.colidx[blk])
    # every triple once
    key = (ccell.astype(np.int64) * nloc + a) * nloc + b
    assert np.array_equal(np.sort(key), np.arange(key.size))
    # fixed order: cells ascending inside a block
    same = blk[1:] == blk[:-1]
    assert (ccell[1:][same] >= ccell[:-1][same]).all()


def test_the_references_own_channel_mesh():
    """data/meshes/bfs3d_coarse60.msh (data shipped with the reference's bfs3d example, bfs3d.py:13-16): read as
    written by gmsh -- 299 nodes, 912 tetrahedra, the channel's volume, every boundary facet tagged, tags 1 / 3 exactly on the
    geometric Dirichlet boundary and tag 2 exactly on the outflow plane (bfs3d.py:23-26) -- and the Scott-Vogelius set-up
    runs on it: Alfeld split, macro-star patches with interiors and skeleton, condensable groups."""
    import os
    from alfi_amd.mesh import read_gmsh
    from alfi_amd.problem import ThreeDimBackwardsFacingStepProblem
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "meshes", "bfs3d_coarse60.msh")
    r = read_gmsh(path)
    assert r.num_vertices == 299 and r.num_cells == 912
    vol = r.cell_geometry()[1]
    assert (vol > 0).all() and np.isclose(vol.sum(), 19.0)
    assert len(r.boundary_tags) == len(r.boundary_facets)
    cent = {k: r.coords[list(k)].mean(axis=0) for k in r.boundary_tags}
    assert all((t == 2) == (abs(cent[k][0] - 10.0) < 1e-9) for k, t in r.boundary_tags.items())
    assert all((t == 1) == (abs(cent[k][0]) < 1e-9) for k, t in r.boundary_tags.items())
    prob = ThreeDimBackwardsFacingStepProblem(1, msh=path)
    m = prob.mesh()                                      # runs check_boundary_tags
    assert m.num_cells == 912
    from alfi_amd.sv import build_sv_hierarchy
    lv, tr = build_sv_hierarchy(prob, 1, 2, Re=100.0)    # [P2]^3: small enough for the CPU suite
    L = lv[-1]
    assert len(lv) == 2 and L.patch_groups is not None
    assert (L.patch_groups >= 0).any() and (L.patch_groups < 0).any()
    assert np.diff(L.patch_ptr).max() > 500
