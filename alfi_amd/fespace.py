"""Nodal velocity function spaces on a mesh hierarchy: dof numbering, Dirichlet sets, star patches, and the sparse
matrices behind the grid transfers.

Roles in the reference (all host-side, setup-time):

* vertex-star patch construction: alfi/relaxation.py:33-34, 110-160 (``Star``) and PETSc's built-in
  ``-pc_patch_construct_type star -pc_patch_construct_dim 0`` (alfi/solver.py:337-338);
* nodal prolongation / its transpose: ``firedrake.prolong / restrict`` (alfi/transfer.py:284-290);
* flux-preserving P1+FB transfer: alfi/bubble.py:204-265 (here collapsed into one sparse matrix);
* coarse-skeleton Dirichlet set and coarse-cell interior blocks of the Schoeberl transfer:
  alfi/transfer.py:121-158 and 13-46.

Vector dofs are node-major, component-minor (bubble.py:86): dof = node * dim + component.
"""
import numpy as np
import scipy.sparse as sp

from .mesh import TET_FACES, child_barycentric, coarse_support_size


def _spread_bits(x, dim):
    x = x.astype(np.uint64)
    if dim == 2:
        x &= np.uint64(0xFFFFFFFF)
        x = (x | (x << np.uint64(16))) & np.uint64(0x0000FFFF0000FFFF)
        x = (x | (x << np.uint64(8))) & np.uint64(0x00FF00FF00FF00FF)
        x = (x | (x << np.uint64(4))) & np.uint64(0x0F0F0F0F0F0F0F0F)
        x = (x | (x << np.uint64(2))) & np.uint64(0x3333333333333333)
        x = (x | (x << np.uint64(1))) & np.uint64(0x5555555555555555)
    else:
        x &= np.uint64(0x1FFFFF)
        x = (x | (x << np.uint64(32))) & np.uint64(0x1F00000000FFFF)
        x = (x | (x << np.uint64(16))) & np.uint64(0x1F0000FF0000FF)
        x = (x | (x << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
        x = (x | (x << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
        x = (x | (x << np.uint64(2))) & np.uint64(0x1249249249249249)
    return x


def morton_order(points):
    """Permutation sorting points along a Z-order curve (stable)."""
    dim = points.shape[1]
    lo, hi = points.min(axis=0), points.max(axis=0)
    span = np.where(hi > lo, hi - lo, 1.0)
    q = np.floor((points - lo) / span * (2 ** 20 - 1) + 0.5).astype(np.int64)
    code = np.zeros(points.shape[0], dtype=np.uint64)
    for a in range(dim):
        code |= _spread_bits(q[:, a], dim) << np.uint64(a)
    return np.argsort(code, kind="stable")


class VectorFunctionSpace(object):
    """[element]^dim on a SimplexMesh.  Scalar nodes are numbered along a Morton curve through their positions so
    that patches, matrix rows and mesh partitions are contiguous-ish in memory (an MI355X choice: x-gathers of the
    patch smoother and the SpMV then hit L2 / Infinity Cache instead of HBM)."""

    def __init__(self, mesh, element, reorder=True, dirichlet=None):
        self.mesh, self.element = mesh, element
        self.dim = mesh.dim
        nv, ne, nf = mesh.num_vertices, mesh.num_edges, mesh.num_faces
        npe = getattr(element, "nodes_per_edge", 1 if element.has_edge_nodes else 0)
        self.nodes_per_edge = npe
        sizes = [nv, ne * npe, nf if element.has_face_nodes else 0]
        self.raw_offsets = np.concatenate([[0], np.cumsum(sizes)])
        self.num_nodes = int(self.raw_offsets[-1])
        pos = [mesh.coords]
        if element.has_edge_nodes:
            # node s of a global edge (v0 < v1) sits at v0 + (s + 1) / (npe + 1) (v1 - v0); raw number edge * npe + s
            x0, x1 = mesh.coords[mesh.edges[:, 0]], mesh.coords[mesh.edges[:, 1]]
            t = (np.arange(npe) + 1.0) / (npe + 1.0)
            pos.append((x0[:, None, :] + t[None, :, None] * (x1 - x0)[:, None, :]).reshape(-1, mesh.dim))
        if element.has_face_nodes:
            pos.append(mesh.coords[mesh.faces].mean(axis=1))
        pos = np.concatenate(pos)
        if reorder:
            order = morton_order(pos)                      # new -> raw
            self.raw2new = np.empty(self.num_nodes, dtype=np.int32)
            self.raw2new[order] = np.arange(self.num_nodes, dtype=np.int32)
            self.node_coords = pos[order]
        else:
            self.raw2new = np.arange(self.num_nodes, dtype=np.int32)
            self.node_coords = pos
        # cell -> node map
        cols = []
        for (edim, loc, sub) in element.entity_nodes:
            if edim == 0:
                cols.append(mesh.cells[:, loc])
            elif edim == 1:
                # the element counts sub from local vertex a to b, the space from the smaller to the larger global vertex
                a, b = element.local_edges[loc]
                same = mesh.cells[:, a] < mesh.cells[:, b]
                s_glob = np.where(same, sub, npe - 1 - sub)
                cols.append(self.raw_offsets[1] + mesh.cell_edges[:, loc].astype(np.int64) * npe + s_glob)
            else:
                cols.append(self.raw_offsets[2] + mesh.cell_faces[:, loc])
        self.cell_nodes = np.ascontiguousarray(self.raw2new[np.stack(cols, axis=1)], dtype=np.int32)
        self.vertex_nodes = self.raw2new[:nv]
        # edge_nodes: (ne,) for one node per edge, (ne, npe) otherwise
        self.edge_nodes = None
        if element.has_edge_nodes:
            en = self.raw2new[self.raw_offsets[1]:self.raw_offsets[2]]
            self.edge_nodes = en if npe == 1 else en.reshape(ne, npe)
        self.face_nodes = self.raw2new[self.raw_offsets[2]:self.raw_offsets[3]] if element.has_face_nodes else None
        # Dirichlet nodes: the whole boundary (ldc2d.py:22-25, ldc3d.py:17-20) unless ``dirichlet`` (a callable on the
        # boundary facet centroids) selects a part of it (bfs3d.py:23-26: inflow and walls, the outflow stays natural)
        vm, em, fm = mesh.boundary_entities(dirichlet)
        bc = [self.vertex_nodes[vm]]
        if element.has_edge_nodes:
            bc.append(np.asarray(self.edge_nodes[em]).ravel())
        if element.has_face_nodes:
            bc.append(self.face_nodes[fm])
        self.bc_nodes = np.unique(np.concatenate(bc)).astype(np.int32)
        self.bc_node_mask = np.zeros(self.num_nodes, dtype=bool)
        self.bc_node_mask[self.bc_nodes] = True

    @property
    def num_dofs(self):
        return self.num_nodes * self.dim

    @property
    def bc_dofs(self):
        return (self.bc_nodes[:, None] * self.dim + np.arange(self.dim, dtype=np.int32)).ravel().astype(np.int32)

    def node_dofs(self, nodes):
        return (np.asarray(nodes)[..., None] * self.dim + np.arange(self.dim)).reshape(*np.shape(nodes)[:-1], -1)

    # -- vertex-star patches ------------------------------------------------------------------------------------
    def star_patches(self, seeds=None):
        """Vertex-star patches (relaxation.py:33-34: transitive support closure of the vertex; the dofs PCPATCH keeps
        are those on the vertex and on the edges/faces containing it, minus Dirichlet dofs).

        Returns (patch_ptr int64 (np+1), patch_dofs int32 sorted ascending within each patch, seed vertex of each
        patch).  Patches are ordered by the node number of their seed vertex; empty patches are dropped."""
        if seeds is None and getattr(self, "_star_cache", None) is not None:
            return self._star_cache               # (the generator and the PatchPC front end both ask: 0.5 s each at config 4)
        m = self.mesh
        nv = m.num_vertices
        seed, node = [np.arange(nv, dtype=np.int64)], [self.vertex_nodes.astype(np.int64)]
        if self.element.has_edge_nodes:
            en = np.asarray(self.edge_nodes, dtype=np.int64).reshape(m.num_edges, -1)
            for j in range(2):
                for s_ in range(en.shape[1]):
                    seed.append(m.edges[:, j].astype(np.int64))
                    node.append(en[:, s_])
        if self.element.has_face_nodes:
            for j in range(3):
                seed.append(m.faces[:, j].astype(np.int64))
                node.append(self.face_nodes.astype(np.int64))
        seed, node = np.concatenate(seed), np.concatenate(node)
        keep = ~self.bc_node_mask[node]
        if seeds is not None:
            sel = np.zeros(nv, dtype=bool)
            sel[np.asarray(seeds)] = True
            keep &= sel[seed]
        seed, node = seed[keep], node[keep]
        # order: patches by the node number of their seed vertex, nodes ascending within a patch -- ONE sort of the packed
        # key (rank, node); both parts are read back out of the sorted key (a lexsort of 10 M pairs cost 3 x as much)
        vnode = self.vertex_nodes.astype(np.int64)
        key = vnode[seed] * np.int64(self.num_nodes) + node
        key.sort()
        rank, node = key // self.num_nodes, key % self.num_nodes
        first = np.ones(rank.shape[0], dtype=bool)
        first[1:] = rank[1:] != rank[:-1]
        start = np.flatnonzero(first)
        counts = np.diff(np.concatenate([start, [rank.shape[0]]]))
        vertex_of_node = np.full(self.num_nodes, -1, dtype=np.int64)
        vertex_of_node[vnode] = np.arange(nv, dtype=np.int64)
        seed = vertex_of_node[rank]
        patch_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64) * self.dim
        patch_dofs = (node[:, None] * self.dim + np.arange(self.dim)).ravel().astype(np.int32)
        out = (patch_ptr, patch_dofs, seed[start].astype(np.int32))
        if seeds is None:
            self._star_cache = out
        return out


# -- grid transfers ---------------------------------------------------------------------------------------------------
def _representatives(cell_nodes, num_nodes):
    """For every node the first (cell, local index) it appears at."""
    flat = cell_nodes.ravel()
    # written back to front, the assignment that stays is the FIRST occurrence (NumPy assigns repeated indices in order,
    # the last one remains) -- what np.unique(..., return_index=True) returns after a stable sort of 15 M entries
    first = np.full(num_nodes, -1, dtype=np.int64)
    first[flat[::-1]] = np.arange(flat.shape[0] - 1, -1, -1, dtype=np.int64)
    assert first.min() >= 0
    nloc = cell_nodes.shape[1]
    return first // nloc, first % nloc


def nodal_prolongation(Vc, Vf, element_c=None, element_f=None, fine_cell_nodes=None, coarse_cell_nodes=None,
                       nfine=None, ncoarse=None):
    """Scalar nodal interpolation matrix (fine nodes x coarse nodes): row of a fine node = values of the coarse basis
    functions at that node (what ``firedrake.prolong`` evaluates, transfer.py:284-286).  The optional arguments allow
    interpolating between sub-spaces (P1 part, bubble part) for the bubble transfer."""
    mf = Vf.mesh
    ec = element_c or Vc.element
    ef = element_f or Vf.element
    fcn = Vf.cell_nodes if fine_cell_nodes is None else fine_cell_nodes
    ccn = Vc.cell_nodes if coarse_cell_nodes is None else coarse_cell_nodes
    nfine = Vf.num_nodes if nfine is None else nfine
    ncoarse = Vc.num_nodes if ncoarse is None else ncoarse
    B = child_barycentric(mf.dim)
    # local interpolation matrices per child type
    Ploc = np.stack([ec.tabulate(ef.node_bary @ B[k])[0] for k in range(B.shape[0])])   # (nch, nloc_f, nloc_c)
    Ploc[np.abs(Ploc) < 1e-14] = 0.0
    cell, loc = _representatives(fcn, nfine)
    k = mf.child_index[cell]
    rows_val = Ploc[k, loc, :]                                      # (nfine, nloc_c)
    cols = ccn[mf.parent_cell[cell]]                                # (nfine, nloc_c)
    # one row per fine node, its entries on distinct coarse nodes: the CSR arrays directly (no COO pass, no duplicates)
    nz = rows_val != 0.0
    indptr = np.concatenate([[0], np.cumsum(nz.sum(axis=1))])
    itype = np.int32 if max(indptr[-1], ncoarse) < 2 ** 31 else np.int64
    P = sp.csr_matrix((rows_val[nz], cols[nz].astype(itype), indptr.astype(itype)), shape=(nfine, ncoarse))
    P.sort_indices()
    return P


def _facet_normals(mesh):
    """Unit normals of all faces of a 3-D mesh (sign irrelevant: only n n^T is used)."""
    x = mesh.coords[mesh.faces]
    n = np.cross(x[:, 1] - x[:, 0], x[:, 2] - x[:, 0])
    return n / np.linalg.norm(n, axis=1)[:, None]


def bubble_prolongation(Vc, Vf):
    """Vector prolongation for [P1+FB]^3 that preserves the flux through coarse facets (alfi/bubble.py:233-265):

        P = C_f . blkdiag(P_P1, P_FB) . S_n . Sp_c

    Sp_c / C_f: nodal <-> hierarchical change of basis (split / combine kernels, bubble.py:57-91, 125-147; the
    divide-by-multiplicity steps average identical per-cell contributions and are the identity in matrix form);
    S_n: b_F <- b_F + 0.6 (b_F . n_F) n_F on every coarse facet (the 1/0.625 rescaling of the normal bubble component,
    bubble.py:29-39, 246-253).  Returns a (3 nf x 3 nc) sparse matrix in the spaces' dof numbering."""
    assert Vc.dim == 3 and Vc.element.bubble and Vc.element.degree == 1
    mc, mf = Vc.mesh, Vf.mesh
    I3 = sp.identity(3, format="csr")

    def split_combine(V):
        m = V.mesh
        nv, nf = m.num_vertices, m.num_faces
        n = V.num_nodes
        vn, fn = V.vertex_nodes.astype(np.int64), V.face_nodes.astype(np.int64)
        # hierarchical numbering: [vertices | faces] in raw mesh order
        rows = np.concatenate([np.arange(nv), nv + np.arange(nf), np.repeat(nv + np.arange(nf), 3)])
        cols = np.concatenate([vn, fn, vn[m.faces].ravel()])
        third = np.full(3 * nf, 1.0 / 3.0)
        split = sp.csr_matrix((np.concatenate([np.ones(nv + nf), -third]), (rows, cols)), shape=(nv + nf, n))
        crow = np.concatenate([vn, fn, np.repeat(fn, 3)])
        ccol = np.concatenate([np.arange(nv), nv + np.arange(nf), m.faces.ravel()])
        combine = sp.csr_matrix((np.concatenate([np.ones(nv + nf), third]), (crow, ccol)), shape=(n, nv + nf))
        return split, combine

    split_c, _ = split_combine(Vc)
    _, combine_f = split_combine(Vf)
    # S_n on the coarse hierarchical space
    nvc, nfc = mc.num_vertices, mc.num_faces
    nrm = _facet_normals(mc)
    blocks = np.eye(3)[None] + 0.6 * nrm[:, :, None] * nrm[:, None, :]
    Sn = sp.block_diag([sp.identity(3 * nvc, format="csr"),
                        sp.bsr_matrix((blocks, np.arange(nfc), np.arange(nfc + 1)), shape=(3 * nfc, 3 * nfc))],
                       format="csr")
    # P1 and FB parts in raw mesh numbering
    from .elements import NodalElement
    p1 = NodalElement(3, 1, False)
    P_p1 = nodal_prolongation(Vc, Vf, p1, p1, mf.cells, mc.cells, mf.num_vertices, mc.num_vertices)

    class _FB(object):          # the bare bubble space: dofs = evaluation at face barycentres, basis = beta_m
        node_bary = Vc.element.node_bary[-4:]

        @staticmethod
        def tabulate(lam):
            from .elements import NodalElement as NE
            return NE._bubbles(np.atleast_2d(lam))

    P_fb = nodal_prolongation(Vc, Vf, _FB, _FB, mf.cell_faces, mc.cell_faces, mf.num_faces, mc.num_faces)
    P_h = sp.block_diag([P_p1, P_fb], format="csr")
    P = sp.kron(combine_f, I3, format="csr") @ sp.kron(P_h, I3, format="csr") @ Sn @ sp.kron(split_c, I3, format="csr")
    P = P.tocsr()
    P.data[np.abs(P.data) < 1e-14] = 0.0
    P.eliminate_zeros()
    P.sort_indices()
    return P


def vector_prolongation(Vc, Vf):
    """The 'standard transfer' of PkP0SchoeberlTransfer (transfer.py:334-356): bubble transfer for 3-D 'CG1'
    elements, plain nodal interpolation otherwise.  Returns CSR (fine dofs x coarse dofs)."""
    if Vc.dim == 3 and Vc.element.bubble and Vc.element.degree == 1:
        return bubble_prolongation(Vc, Vf)
    P = nodal_prolongation(Vc, Vf)
    return sp.kron(P, sp.identity(Vc.dim, format="csr"), format="csr")


def skeleton_node_mask(Vf):
    """True for fine nodes on the closure of a coarse facet: the Dirichlet set of transfer.py:121-158."""
    m = Vf.mesh
    assert m.vertex_parents is not None, "not a refined mesh"
    mask = np.ones(Vf.num_nodes, dtype=bool)        # vertices: always on the skeleton
    if Vf.element.has_edge_nodes:
        mask[Vf.edge_nodes] = coarse_support_size(m, m.edges) <= m.dim
    if Vf.element.has_face_nodes:
        mask[Vf.face_nodes] = coarse_support_size(m, m.faces) <= m.dim
    return mask


def coarse_cell_blocks(Vf):
    """Interior node blocks of the Schoeberl transfer: one block per coarse cell holding the fine nodes in the closure
    of its children that are not on the coarse skeleton (transfer.py:13-46 with PatchPC dropping the Dirichlet nodes).
    Returns (num_coarse_cells, m) int32 node numbers, ascending within each block."""
    m = Vf.mesh
    skel = skeleton_node_mask(Vf)
    interior = ~skel[Vf.cell_nodes]                                  # (nfc, nloc)
    ci, li = np.nonzero(interior)
    pairs = np.stack([m.parent_cell[ci].astype(np.int64), Vf.cell_nodes[ci, li].astype(np.int64)], axis=1)
    key = np.unique(pairs[:, 0] * Vf.num_nodes + pairs[:, 1])
    parent, node = key // Vf.num_nodes, key % Vf.num_nodes
    counts = np.bincount(parent)
    assert counts.min() == counts.max(), "non-uniform interior blocks"
    return node.reshape(-1, counts[0]).astype(np.int32)


def injection_map(Vc, Vf):
    """Fine node carrying each coarse node: the velocity dofs are point evaluations and every
    coarse node -- vertex, edge midpoint, face barycentre -- is also a node of the once-refined space (coarse edge
    midpoints become fine vertices, the central child of a coarse face has the coarse face's barycentre), so Firedrake's
    ``inject`` for these spaces (the third entry of the transfer tuple, alfi/solver.py:595; used to move the Newton state
    to the coarse levels, alfi/stabilisation.py:41) is the index map returned here.  (num coarse nodes,) int32."""
    scale = 3.0 * 2.0 ** 12    # integer lattice fine enough to separate nodes (spacing >= span / (6 N)), 3 axes fit int64
    lo = Vf.node_coords.min(axis=0)
    span = (Vf.node_coords.max(axis=0) - lo).max()

    def key(x):
        q = np.rint((x - lo) / span * scale).astype(np.int64)
        k = q[:, 0]
        for a in range(1, x.shape[1]):
            k = k * np.int64(4 * scale) + q[:, a]
        return k
    kf, kc = key(Vf.node_coords), key(Vc.node_coords)
    order = np.argsort(kf)
    pos = np.searchsorted(kf[order], kc)
    pos[pos >= len(kf)] = len(kf) - 1
    out = order[pos]
    if not np.array_equal(kf[out], kc):
        raise ValueError("a coarse node is not a node of the fine space: inject is not a nodal map for this element")
    return out.astype(np.int32)
