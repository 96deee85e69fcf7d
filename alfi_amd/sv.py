"""Scott-Vogelius side of the reference (towards BASELINE.json config 5): [P_k]^d velocities on barycentrically refined
meshes with the FULL grad-div term, macro-star patches and the macro-cell Schoeberl transfer.

Reference: ``ScottVogeliusSolver`` (alfi/solver.py:604-662: forms with ``gamma * inner(div(u), div(v))``, no cell_avg;
``patch = macro``), ``BaryMeshHierarchy`` (alfi/bary.py:29-194: level l = Alfeld split of the l-times refined base mesh;
the levels are not nested, only their macro meshes are), ``SVSchoeberlTransfer`` (alfi/transfer.py:293-309) with
``CoarseCellMacroPatches`` (transfer.py:49-88: one interior block per coarse MACRO cell) and Firedrake's non-nested
``prolong`` as the standard transfer (transfer.py:284-290).

Everything here is host-side input generation (the role Firedrake plays for the reference); the arithmetic of smoother,
transfers and cycles is the same libalfi_hip.so entry points as for the PkP0 discretisation.  Implemented for k = 2 and,
in 3-D, k = 3 (the inf-sup stable pair of config 5: macro stars of up to ~1600 dofs, macro-cell transfer blocks of 390).
"""
import os

import numpy as np
import scipy.sparse as sp

from . import _hostlib
from .elements import NodalElement
from .fespace import VectorFunctionSpace
from .mesh import bary_refine, mesh_hierarchy
from .problem import BSR, LevelData, TransferData
from .relaxation import MacroStar, PlexLike, patch_points_to_dofs


def macro_labels(mesh):
    """``MacroVertices`` = 1 on the vertices of the mesh that was split (bary.py:18-19), in PlexLike point numbering."""
    return {"MacroVertices": {int(mesh.num_cells + v): 1 for v in np.flatnonzero(mesh.macro_vertex_mask)}}


def macro_star_patches(V):
    """The reference's MacroStar constructor (relaxation.py:163-177), literally, on the Alfeld-split mesh of V."""
    dm = PlexLike(V.mesh, labels=macro_labels(V.mesh))

    class _PC(object):
        options = {}

        def getDM(self):
            return dm

        def getOptionsPrefix(self):
            return ""
    ms = MacroStar()
    patches, iterset = ms(_PC())
    ptr, dofs, kept = patch_points_to_dofs(V, dm, patches)
    seeds = np.array([ms.seeds[i] - dm.vStart for i in kept], dtype=np.int64)     # the macro vertex of every patch
    return ptr, dofs, seeds


def macro_star_patches_fast(V):
    """The same patches as ``macro_star_patches`` (the reference's MacroStar constructor, relaxation.py:163-177, with its
    literal label test on EVERY closure point) from sparse incidence products instead of one Python closure walk per
    point: 19 of the 34 s of host generation at config 5's size, 150 of 281 s with one more refinement.

    For a macro vertex v let C(v) = closure of the cells holding v.  The constructor returns star(v) and the stars of all
    points of C(v) that are not macro vertices, i.e. in terms of the entities that carry nodes
      vertices:  v and the non-macro vertices (macro-cell barycentres) of C(v);
      edges:     the edges of C(v) and the edges holding one of those barycentres;
      faces:     the faces of C(v), the faces holding an edge of C(v), the faces holding one of those barycentres.
    tests/test_sv.py checks equality with the literal constructor patch by patch (2-D and 3-D)."""
    import scipy.sparse as sp
    mesh, d = V.mesh, V.dim
    nc, nv, ne, nf = mesh.num_cells, mesh.num_vertices, mesh.num_edges, mesh.num_faces
    macro = np.flatnonzero(mesh.macro_vertex_mask)

    def inc(rows, cols, shape):
        return sp.csr_matrix((np.ones(rows.size, dtype=np.int32), (rows, cols)), shape=shape)
    cells = np.repeat(np.arange(nc), d + 1)
    Mvc = inc(mesh.cells.ravel(), cells, (nv, nc))                                    # vertex x cell
    Mec = inc(mesh.cell_edges.ravel(), np.repeat(np.arange(nc), mesh.cell_edges.shape[1]), (ne, nc))
    Mev = inc(np.repeat(np.arange(ne), 2), mesh.edges.ravel(), (ne, nv))              # edge x vertex
    Cv = Mvc.T.tocsr()[:, macro]                                                      # cell x macro vertex: cells holding v
    Vc = (Mvc @ Cv).tocsr()                                                           # vertices of C(v)
    nonmacro = sp.diags((~mesh.macro_vertex_mask).astype(np.int32))
    Bv = (nonmacro @ Vc).tocsr()                                                      # ... that are not macro vertices
    Ec = (Mec @ Cv).tocsr()
    E = (Ec + Mev @ Bv).tocsc()
    Bc = Bv.tocsc()
    F = None
    if d == 3 and V.element.has_face_nodes:
        Mfc = inc(mesh.cell_faces.ravel(), np.repeat(np.arange(nc), 4), (nf, nc))
        Mfv = inc(np.repeat(np.arange(nf), 3), mesh.faces.ravel(), (nf, nv))
        # face x edge: the three vertex pairs of a face looked up among the (sorted) edges
        ekey = mesh.edges[:, 0].astype(np.int64) * nv + mesh.edges[:, 1]
        order = np.argsort(ekey)
        f = mesh.faces.astype(np.int64)
        fe = []
        for a, b in ((0, 1), (0, 2), (1, 2)):
            lo, hi = np.minimum(f[:, a], f[:, b]), np.maximum(f[:, a], f[:, b])
            fe.append(order[np.searchsorted(ekey[order], lo * nv + hi)])
        Mfe = inc(np.repeat(np.arange(nf), 3), np.stack(fe, axis=1).ravel(), (nf, ne))
        F = (Mfc @ Cv + Mfe @ Ec + Mfv @ Bv).tocsc()
    en = None if V.edge_nodes is None else np.asarray(V.edge_nodes, dtype=np.int64).reshape(ne, -1)
    ptr, dofs, seeds = [0], [], []
    comp = np.arange(d, dtype=np.int64)
    for j, v in enumerate(macro):
        nodes = [np.array([V.vertex_nodes[v]], dtype=np.int64),
                 V.vertex_nodes[Bc.indices[Bc.indptr[j]:Bc.indptr[j + 1]]].astype(np.int64)]
        if en is not None:
            nodes.append(en[E.indices[E.indptr[j]:E.indptr[j + 1]]].ravel())
        if F is not None:
            nodes.append(V.face_nodes[F.indices[F.indptr[j]:F.indptr[j + 1]]].astype(np.int64))
        nodes = np.unique(np.concatenate(nodes))
        nodes = nodes[~V.bc_node_mask[nodes]]
        if nodes.size == 0:
            continue
        dofs.append((nodes[:, None] * d + comp).ravel())
        ptr.append(ptr[-1] + nodes.size * d)
        seeds.append(v)
    return (np.array(ptr, dtype=np.int64), np.concatenate(dofs).astype(np.int32) if dofs else np.zeros(0, dtype=np.int32),
            np.array(seeds, dtype=np.int64))


def macro_cell_groups(V, patch_dofs):
    """Group labels for condensed patch factors (alfi_patches_set_groups / hip.Level.set_patch_groups): a dof whose node
    lies strictly inside one macro cell of the Alfeld-split mesh -- every cell holding the node is a child of the same macro
    cell (bary cell c*(d+1)+i is child i of macro cell c, bary.py:148-151) -- gets that macro cell's number, every dof on the
    macro skeleton gets -1.  Interiors of different macro cells never share a cell, hence no operator entry: inside any
    patch they are coupled only through the skeleton."""
    mesh, d = V.mesh, V.dim
    nloc = V.cell_nodes.shape[1]
    macro = np.repeat(np.arange(mesh.num_cells, dtype=np.int64) // (d + 1), nloc)
    lo = np.full(V.num_nodes, np.iinfo(np.int64).max, dtype=np.int64)
    hi = np.full(V.num_nodes, -1, dtype=np.int64)
    np.minimum.at(lo, V.cell_nodes.ravel(), macro)
    np.maximum.at(hi, V.cell_nodes.ravel(), macro)
    label = np.where(lo == hi, lo, -1)
    return label[np.asarray(patch_dofs, dtype=np.int64) // d].astype(np.int32)


def _coarse_macro_cell_of_nodes(Vf):
    """For every fine node one fine cell holding it, and the coarse MACRO cell that cell lies in."""
    mf = Vf.mesh
    d = mf.dim
    flat = Vf.cell_nodes.ravel()
    _, first = np.unique(flat, return_index=True)
    cell = first // Vf.cell_nodes.shape[1]
    fine_macro = cell // (d + 1)                                  # bary cell c*(d+1)+i is a child of macro cell c
    return mf.macro_mesh.parent_cell[fine_macro].astype(np.int64)


def _barycentric(mesh, cells, x):
    """Barycentric coordinates of the points x (m, dim) with respect to the cells ``cells`` (m,) of mesh."""
    g, _ = mesh.cell_geometry()
    v0 = mesh.coords[mesh.cells[cells, 0]]
    lam = np.einsum("mix,mx->mi", g[cells], x - v0)
    lam[:, 0] += 1.0
    return lam


def bary_prolongation(Vc, Vf):
    """Nodal interpolation between the (non-nested) Alfeld levels: every fine node is located in one of the d+1 sub-cells
    of the coarse macro cell it lies in, and the coarse basis is evaluated there -- what firedrake.prolong does on a
    non-nested hierarchy with the coarse_to_fine maps of bary.py:137-160.  Scalar CSR (fine nodes x coarse nodes)."""
    mc = Vc.mesh
    d = mc.dim
    C = _coarse_macro_cell_of_nodes(Vf)
    x = Vf.node_coords
    best, bestlam = None, None
    for i in range(d + 1):
        cand = C * (d + 1) + i
        lam = _barycentric(mc, cand, x)
        score = lam.min(axis=1)
        if best is None:
            best, bestlam, bestscore = cand.copy(), lam, score
        else:
            better = score > bestscore
            best[better], bestlam[better], bestscore[better] = cand[better], lam[better], score[better]
    assert bestscore.min() > -1e-9, "a fine node was not located in its coarse macro cell"
    phi = Vc.element.tabulate(np.clip(bestlam, 0.0, 1.0))[0]        # (nfine, nloc)
    phi[np.abs(phi) < 1e-13] = 0.0
    rows = np.repeat(np.arange(Vf.num_nodes), phi.shape[1])
    cols = Vc.cell_nodes[best].ravel()
    P = sp.csr_matrix((phi.ravel(), (rows, cols)), shape=(Vf.num_nodes, Vc.num_nodes))
    P.sum_duplicates()
    P.eliminate_zeros()
    P.sort_indices()
    return P


def bary_injection(Vc, Vf):
    """firedrake ``inject`` on the non-nested bary hierarchy for the nodal velocity spaces (solver.py:641-644): the fine
    function evaluated at the coarse nodes.  Every coarse node is located in one of the 2^d (d+1) fine cells of the coarse
    macro cell it belongs to.  Scalar CSR (coarse nodes x fine nodes); used to move the Newton state to the coarse levels."""
    mf = Vf.mesh
    d = mf.dim
    nch = 2 ** d
    flat = Vc.cell_nodes.ravel()
    _, first = np.unique(flat, return_index=True)
    C = (first // Vc.cell_nodes.shape[1]) // (d + 1)                # coarse macro cell of every coarse node
    x = Vc.node_coords
    best = bestlam = bestscore = None
    for j in range(nch):
        for i in range(d + 1):
            cand = (C * nch + j) * (d + 1) + i
            lam = _barycentric(mf, cand, x)
            score = lam.min(axis=1)
            if best is None:
                best, bestlam, bestscore = cand.copy(), lam, score
            else:
                better = score > bestscore
                best[better], bestlam[better], bestscore[better] = cand[better], lam[better], score[better]
    assert bestscore.min() > -1e-9, "a coarse node was not located in the fine cells of its macro cell"
    phi = Vf.element.tabulate(np.clip(bestlam, 0.0, 1.0))[0]
    phi[np.abs(phi) < 1e-13] = 0.0
    rows = np.repeat(np.arange(Vc.num_nodes), phi.shape[1])
    J = sp.csr_matrix((phi.ravel(), (rows, Vf.cell_nodes[best].ravel())), shape=(Vc.num_nodes, Vf.num_nodes))
    J.sum_duplicates()
    J.eliminate_zeros()
    J.sort_indices()
    return J


def macro_skeleton_mask(Vf):
    """True for fine nodes on the closure of a coarse MACRO facet: the Dirichlet set of fix_coarse_boundaries
    (transfer.py:121-158) on a bary hierarchy, where the ``prolongation`` label marks the facets of the coarse macro mesh."""
    coarse_macro = Vf.mesh.macro_mesh                                  # refined macro mesh; its parent is the coarse macro mesh
    C = _coarse_macro_cell_of_nodes(Vf)
    # barycentric coordinates with respect to the coarse macro cell: reconstruct that mesh's geometry from the parents
    cm_cells = _coarse_macro_cells(coarse_macro)
    x = Vf.node_coords
    v = coarse_macro.coords[cm_cells[C]]                               # (m, d+1, d) vertices of the coarse macro cell
    T = (v[:, 1:, :] - v[:, :1, :]).transpose(0, 2, 1)
    rhs = x - v[:, 0, :]
    lam_rest = np.linalg.solve(T, rhs[:, :, None])[:, :, 0]
    lam = np.concatenate([1.0 - lam_rest.sum(axis=1, keepdims=True), lam_rest], axis=1)
    return lam.min(axis=1) < 1e-9


def _coarse_macro_cells(refined_macro):
    """Vertex tuples of the cells of the mesh ``refined_macro`` was refined from (child 0..d keep one parent vertex each)."""
    d = refined_macro.dim
    nch = 2 ** d
    ncoarse = refined_macro.num_cells // nch
    out = np.empty((ncoarse, d + 1), dtype=np.int64)
    for i in range(d + 1):
        # child i of coarse cell c is fine cell c*nch + i and its local vertex i is the coarse vertex i (mesh.children_table)
        out[:, i] = refined_macro.cells[np.arange(ncoarse) * nch + i, i]
    return out


def macro_cell_blocks(Vf):
    """transfer.py:49-88 (CoarseCellMacroPatches): one block per coarse macro cell with the fine nodes strictly inside it.
    (num coarse macro cells, m / dim) node numbers, ascending."""
    C = _coarse_macro_cell_of_nodes(Vf)
    interior = ~macro_skeleton_mask(Vf)
    nodes = np.flatnonzero(interior)
    order = np.lexsort((nodes, C[nodes]))
    counts = np.bincount(C[nodes])
    assert counts.min() == counts.max(), "non-uniform macro interior blocks"
    return nodes[order].reshape(-1, counts[0]).astype(np.int32)


def build_sv_transfer_data(Vc, Vf, nu, gamma, graph):
    mesh, d = Vf.mesh, Vf.dim
    el = Vf.element
    rowptr, colidx = graph
    g, vol = mesh.cell_geometry()
    tens = el.reference_tensors()
    T = TransferData()
    blk_nodes = macro_cell_blocks(Vf)
    T.blk_dofs = np.ascontiguousarray(Vf.node_dofs(blk_nodes), dtype=np.int32)
    nch = (2 ** d) * (d + 1)                                         # fine bary cells per coarse macro cell, contiguous
    T.K_II, T.D_II = _hostlib.interior_blocks(Vf.cell_nodes, g, vol, tens, d, blk_nodes, Vf.num_nodes, nch,
                                              full_div=True)
    rows = blk_nodes.ravel().astype(np.int64)
    cnt = (rowptr[rows + 1] - rowptr[rows]).astype(np.int64)
    ptr = np.concatenate([[0], np.cumsum(cnt)])
    idx = np.repeat(rowptr[rows].astype(np.int64) - ptr[:-1], cnt) + np.arange(ptr[-1])
    di_rowptr, di_colidx = ptr.astype(np.int32), colidx[idx]
    row_map = np.full(Vf.num_nodes, -1, dtype=np.int32)
    row_map[rows] = np.arange(rows.shape[0], dtype=np.int32)
    di_vals = _hostlib.assemble_bsr(Vf.cell_nodes, g, vol, tens, d, di_rowptr, di_colidx, gamma_full=1.0,
                                    row_map=row_map)
    T.D_I = BSR(rows.shape[0], Vf.num_nodes, d, di_rowptr, di_colidx, di_vals)
    T.D_IT = T.D_I.transpose()
    Pv = sp.kron(bary_prolongation(Vc, Vf), sp.identity(d, format="csr"), format="csr")
    T.P = BSR.from_scipy(Pv, d)
    T.PT = T.P.transpose()
    T.PT_plain = T.PT
    T.inject_map = None
    T.inject_matrix = bary_injection(Vc, Vf)
    T.nu, T.gamma = nu, gamma
    T.n_f, T.n_c = Vf.num_dofs, Vc.num_dofs
    T.bc_dofs_f, T.bc_dofs_c = Vf.bc_dofs, Vc.bc_dofs
    T.skeleton_dofs = np.flatnonzero(np.repeat(macro_skeleton_mask(Vf), d))
    return T


def build_sv_pressure_coupling(L, zero_bc_columns=True, both=False):
    """The discontinuous P_{k-1} pressure space of ScottVogeliusSolver.function_space (solver.py:624-629) on the level's
    (Alfeld-split) mesh, nodal basis psi per cell: the discrete divergence B[(c, j), (a, x)] = -int_c psi_j d_x phi_a
    (``- div(u) * q * dx``, solver.py:619; Dirichlet velocity columns zeroed for the Jacobian) and the block-diagonal
    pressure mass matrix with its inverse (what DGMassInv applies, solver.py:15-38).  With these the full grad-div term of
    the level operator is gamma B^T M^-1 B (div [P_k]^d is contained in P_{k-1}^dg).  Returns scipy CSR (B, M, Minv);
    ``both``: (B with the Dirichlet columns zeroed, B with all columns, M, Minv) from one pass.

    The CSR arrays are written directly -- every row (c, j) holds the nloc * d velocity dofs of cell c, sorted once per
    cell -- instead of going through a COO triple of 60 entries per pressure dof (5 s of sorting at 440 k velocity dofs,
    twice per solver set-up)."""
    from .elements import simplex_quadrature
    V = L.V
    mesh, d, el = V.mesh, V.dim, V.element
    pel = NodalElement(d, el.degree - 1, False)
    lam, wq = simplex_quadrature(d, 6)
    phi, dphi = el.tabulate(lam)                                    # (q, a), (q, a, i)
    psi = pel.tabulate(lam)[0]                                      # (q, j)
    g, vol = mesh.cell_geometry()
    nc, nloc, npl = mesh.num_cells, el.nloc, pel.nloc
    ref = np.einsum("q,qj,qai->jai", wq, psi, dphi)                 # reference integrals of psi_j d_i phi_a
    Bc = -np.einsum("c,jai,cix->cjax", vol, ref, g).reshape(nc, npl, nloc * d)      # (c, j, (a, x))
    cols = (V.cell_nodes[:, :, None] * d + np.arange(d)).reshape(nc, nloc * d)       # the same for every j of the cell
    order = np.argsort(cols, axis=1, kind="stable")
    cols = np.take_along_axis(cols, order, axis=1)
    Bc = np.take_along_axis(Bc, order[:, None, :], axis=2)
    w = nloc * d
    indptr = np.arange(nc * npl + 1, dtype=np.int64) * w
    indices = np.broadcast_to(cols[:, None, :], (nc, npl, w)).reshape(-1)
    idx_t = np.int32 if indices.size < 2 ** 31 and V.num_dofs < 2 ** 31 else np.int64

    def make(data):
        B = sp.csr_matrix((data.reshape(-1), indices.astype(idx_t), indptr.astype(idx_t)), shape=(nc * npl, V.num_dofs))
        B.eliminate_zeros()
        B.has_sorted_indices = True
        return B
    keep = np.ones(V.num_dofs)
    keep[V.bc_dofs] = 0.0
    mref = np.einsum("q,qj,ql->jl", wq, psi, psi)                   # reference mass matrix (unit volume)
    M = sp.block_diag([mref], format="csr") if nc == 0 else sp.kron(sp.diags(vol), mref, format="csr")
    Minv = sp.kron(sp.diags(1.0 / vol), np.linalg.inv(mref), format="csr")
    if both:
        return make(Bc * keep[cols][:, None, :]), make(Bc), M, Minv
    return make(Bc * keep[cols][:, None, :] if zero_bc_columns else Bc), M, Minv


def build_sv_hierarchy(problem, nref, k, Re, gamma=1e4, advect=True, patches=True):
    """The Scott-Vogelius analogue of ``problem.build_hierarchy``: levels 0..nref on the bary hierarchy."""
    dim = problem.dim
    if k not in (2, 3) or (k == 3 and dim != 3):
        raise NotImplementedError("Scott-Vogelius velocities: [P2]^d, and [P3]^3 (the inf-sup stable 3-D pair of config 5)")
    element = NodalElement(dim, k, False)
    mh = [bary_refine(m) for m in mesh_hierarchy(problem.mesh(), nref)]
    nu = problem.char_length() * problem.char_velocity() / Re if Re > 0 else problem.char_length() * problem.char_velocity()
    adv = 1.0 if (advect and Re > 0) else 0.0
    levels, transfers, Vprev = [], [], None
    for l, mesh in enumerate(mh):
        V = VectorFunctionSpace(mesh, element, dirichlet=getattr(problem, "dirichlet_facets", None))
        d = V.dim
        L = LevelData()
        L.V, L.level, L.n, L.bs = V, l, V.num_dofs, d
        rowptr, colidx = _hostlib.node_graph(V.cell_nodes, V.num_nodes)
        g, vol = mesh.cell_geometry()
        tens = element.reference_tensors()
        wind = problem.driver(V.node_coords)
        A = _hostlib.assemble_bsr(V.cell_nodes, g, vol, tens, d, rowptr, colidx, nu=nu, adv=adv,
                                  wind=wind if adv else None, gamma_full=gamma)
        _hostlib.apply_bc_bsr(V.num_nodes, d, rowptr, colidx, A, np.repeat(V.bc_node_mask, d))
        L.A = BSR(V.num_nodes, V.num_nodes, d, rowptr, colidx, A)
        L.bc_dofs = V.bc_dofs
        L.nu, L.gamma = nu, gamma
        if patches and l > 0:
            # (ALFI_MACROSTAR_LITERAL=1: the constructor's own point-by-point walk instead of the incidence products)
            literal = os.environ.get("ALFI_MACROSTAR_LITERAL") == "1"
            L.patch_ptr, L.patch_dofs, L.patch_seeds = (macro_star_patches if literal else macro_star_patches_fast)(V)
            # the factors of these patches can be stored condensed: interiors of the macro cells + skeleton
            L.patch_groups = macro_cell_groups(V, L.patch_dofs)
        if l > 0:
            transfers.append(build_sv_transfer_data(Vprev, V, nu, gamma, (rowptr, colidx)))
        levels.append(L)
        Vprev = V
    return levels, transfers
