"""Rank-local generation of the multigrid inputs: sparsity now, values of a row subset on demand.

In the reference every MPI rank assembles only its own mesh partition (Firedrake assembles over the cells of the rank's
DMPlex partition with the vertex-star overlap of alfi/solver.py:604-605).  The first version of this package generated
the *global* hierarchy on every rank and cut the rank's rows out of it (26-32 GB of host memory and the whole assembly
time per rank at config 4).  The classes here keep the global part to integers -- meshes, dof numbering, the node
graph, patches, coarse-cell blocks: what the partitioner needs -- and produce operator / transfer VALUES only for the
rows a rank asks for.  They offer the attributes and methods alfi_amd.dist uses on the global containers
(``problem.BSR`` / ``problem.TransferData``), so the partitioner and the localisation run unchanged on either kind.

Every numeric routine is the one the global generator uses (csrc/host_assemble.cpp through _hostlib), called with a row
map; ``tests/test_lazy.py`` checks rank-local against cut-from-global for every rank.
"""
import numpy as np
import scipy.sparse as sp

from . import _hostlib
from .mesh import child_barycentric


def _take_rows(rowptr, colidx, rows):
    """(ptr int64, cols) of the graph rows ``rows``, concatenated in that order."""
    rows = np.asarray(rows, dtype=np.int64)
    cnt = (rowptr[rows + 1] - rowptr[rows]).astype(np.int64)
    ptr = np.concatenate([[0], np.cumsum(cnt)])
    idx = np.repeat(rowptr[rows].astype(np.int64) - ptr[:-1], cnt) + np.arange(ptr[-1])
    return ptr, colidx[idx]


def _row_map(num_nodes, rows):
    m = np.full(num_nodes, -1, dtype=np.int32)
    m[rows] = np.arange(len(rows), dtype=np.int32)
    return m


class LazyOperator(object):
    """Level operator nu K + gamma D + adv N(w) with Dirichlet rows / columns replaced by the identity (what
    ``problem.build_hierarchy`` assembles, alfi/solver.py:565-568): the node graph, and ``select_rows`` assembling the
    block rows of a node subset (columns stay global)."""

    is_lazy = True

    def __init__(self, V, rowptr, colidx, geometry, tensors, nu, gamma, adv, wind, full_div=False, with_bc=True, values=True):
        """full_div: the Scott-Vogelius grad-div term gamma (div u, div v) (solver.py:616) instead of the cell-averaged one.
        with_bc=False: the raw form, Dirichlet rows / columns as assembled (the parts K, D of the device-side refresh).
        values=False: ``select_rows`` delivers the sparsity of the rows only (``vals`` None): the caller forms the values on the
        device (build_hierarchy(operator_values=False))."""
        self.values = bool(values)
        self.full_div = bool(full_div)
        self.with_bc = bool(with_bc)
        self.V = V
        self.nbrows = self.nbcols = V.num_nodes
        self.bs = V.dim
        self.rowptr, self.colidx = rowptr, colidx
        self.g, self.vol = geometry
        self.tens = tensors
        self.nu, self.gamma, self.adv, self.wind = nu, gamma, adv, wind
        self._bcmask = np.repeat(V.bc_node_mask, V.dim)

    @property
    def nnzb(self):
        return self.colidx.shape[0]

    @property
    def shape(self):
        return (self.nbrows * self.bs, self.nbcols * self.bs)

    def select_rows(self, rows):
        from .problem import BSR
        rows = np.asarray(rows, dtype=np.int64)
        V, d = self.V, self.bs
        ptr, cols = _take_rows(self.rowptr, self.colidx, rows)
        ptr32 = ptr.astype(np.int32)
        if not self.values:
            return BSR(len(rows), self.nbcols, d, ptr32, cols, None)
        vals = _hostlib.assemble_bsr(V.cell_nodes, self.g, self.vol, self.tens, d, ptr32, cols, nu=self.nu,
                                     gamma=0.0 if self.full_div else self.gamma,
                                     gamma_full=self.gamma if self.full_div else 0.0, adv=self.adv,
                                     wind=self.wind if self.adv else None, row_map=_row_map(V.num_nodes, rows))
        if self.with_bc:
            _hostlib.apply_bc_bsr(len(rows), d, ptr32, cols, vals, self._bcmask, row_ids=rows)
        return BSR(len(rows), self.nbcols, d, ptr32, cols, vals)

    def materialise(self):
        return self.select_rows(np.arange(self.nbrows))

    def to_scipy(self):
        return self.materialise().to_scipy()


class LazyInteriorRows(object):
    """Rows of the cell-averaged grad-div matrix (gamma = 1, no boundary conditions; ``bform`` of alfi/transfer.py:326-332)
    for the coarse-cell interior nodes, in block order: row r belongs to fine node ``row_nodes[r]`` and has that node's
    graph row as its sparsity."""

    is_lazy = True

    def __init__(self, V, rowptr, colidx, geometry, tensors, blk_nodes):
        self.V = V
        self.bs = V.dim
        self.row_nodes = np.ascontiguousarray(blk_nodes, dtype=np.int64).ravel()
        self.mb = blk_nodes.shape[1]
        self.nbrows, self.nbcols = self.row_nodes.shape[0], V.num_nodes
        self._rowptr, self._colidx = rowptr, colidx
        self.g, self.vol = geometry
        self.tens = tensors

    def nodes_with_cols_in(self, lo, hi):
        """Mask over ALL nodes: graph row holds a column in [lo, hi).  The node graph is symmetric (two nodes are coupled
        iff they share a cell), so these are the columns of the rows lo .. hi-1: work proportional to the range."""
        mask = np.zeros(self.nbcols, dtype=bool)
        mask[self._colidx[self._rowptr[lo]:self._rowptr[hi]]] = True
        return mask

    def blocks_with_cols_in(self, lo, hi):
        key = (int(lo), int(hi))
        if getattr(self, "_bw_key", None) != key:
            self._bw_key = key
            self._bw = self.nodes_with_cols_in(lo, hi)[self.row_nodes].reshape(-1, self.mb).any(axis=1)
        return self._bw

    def cols_of_rows(self, rows):
        return _take_rows(self._rowptr, self._colidx, self.row_nodes[np.asarray(rows, dtype=np.int64)])[1]

    def select_rows(self, rows):
        from .problem import BSR
        nodes = self.row_nodes[np.asarray(rows, dtype=np.int64)]
        ptr, cols = _take_rows(self._rowptr, self._colidx, nodes)
        ptr32 = ptr.astype(np.int32)
        vals = _hostlib.assemble_bsr(self.V.cell_nodes, self.g, self.vol, self.tens, self.bs, ptr32, cols, gamma=1.0,
                                     row_map=_row_map(self.V.num_nodes, nodes))
        return BSR(len(nodes), self.nbcols, self.bs, ptr32, cols, vals)


class LazyProlongation(object):
    """Standard prolongation of the Schoeberl transfer (fine nodes x coarse nodes, bs x bs blocks): nodal interpolation
    (``firedrake.prolong``, alfi/transfer.py:284-290), or -- ``bubble`` -- the flux-preserving [P1+FB]^3 transfer of
    alfi/bubble.py:233-265 in the matrix form of ``fespace.bubble_prolongation``; rows of a fine-node subset on demand."""

    is_lazy = True

    def __init__(self, Vc, Vf, bubble=False):
        self.Vc, self.Vf, self.bubble = Vc, Vf, bubble
        self.bs = Vf.dim
        self.nbrows, self.nbcols = Vf.num_nodes, Vc.num_nodes
        self._reps = None
        self._bub = None

    # -- scalar nodal interpolation rows ---------------------------------------------------------------------------
    def _nodal_rows(self, nodes, fcn=None, ccn=None, ec=None, ef=None, nfine=None, ncoarse=None, key="n"):
        Vc, Vf = self.Vc, self.Vf
        mf = Vf.mesh
        ec, ef = ec or Vc.element, ef or Vf.element
        fcn = Vf.cell_nodes if fcn is None else fcn
        ccn = Vc.cell_nodes if ccn is None else ccn
        nfine = Vf.num_nodes if nfine is None else nfine
        ncoarse = Vc.num_nodes if ncoarse is None else ncoarse
        if self._reps is None:
            self._reps = {}
        if key not in self._reps:
            from .fespace import _representatives
            self._reps[key] = _representatives(fcn, nfine)
        cell, loc = self._reps[key]
        nodes = np.asarray(nodes, dtype=np.int64)
        cell, loc = cell[nodes], loc[nodes]
        B = child_barycentric(mf.dim)
        Ploc = np.stack([ec.tabulate(ef.node_bary @ B[k])[0] for k in range(B.shape[0])])
        Ploc[np.abs(Ploc) < 1e-14] = 0.0
        vals = Ploc[mf.child_index[cell], loc, :]
        cols = ccn[mf.parent_cell[cell]]
        # one row per fine node, its entries on distinct coarse nodes: the CSR arrays directly (fespace.nodal_prolongation)
        nz = vals != 0.0
        indptr = np.concatenate([[0], np.cumsum(nz.sum(axis=1))])
        itype = np.int32 if max(indptr[-1], ncoarse) < 2 ** 31 else np.int64
        P = sp.csr_matrix((vals[nz], cols[nz].astype(itype), indptr.astype(itype)), shape=(len(nodes), ncoarse))
        P.sort_indices()
        return P

    def _bubble_rows(self, nodes):
        """Rows ``nodes`` of P = C_f . blkdiag(P_P1, P_FB) . S_n . Sp_c, multiplied left to right like the global product."""
        from .elements import NodalElement
        from .fespace import _facet_normals
        Vc, Vf = self.Vc, self.Vf
        mc, mf = Vc.mesh, Vf.mesh
        I3 = sp.identity(3, format="csr")
        if self._bub is None:
            nvc, nfc = mc.num_vertices, mc.num_faces
            vn, fn = Vc.vertex_nodes.astype(np.int64), Vc.face_nodes.astype(np.int64)
            rows = np.concatenate([np.arange(nvc), nvc + np.arange(nfc), np.repeat(nvc + np.arange(nfc), 3)])
            cols = np.concatenate([vn, fn, vn[mc.faces].ravel()])
            third = np.full(3 * nfc, 1.0 / 3.0)
            split_c = sp.csr_matrix((np.concatenate([np.ones(nvc + nfc), -third]), (rows, cols)),
                                    shape=(nvc + nfc, Vc.num_nodes))
            nrm = _facet_normals(mc)
            blocks = np.eye(3)[None] + 0.6 * nrm[:, :, None] * nrm[:, None, :]
            Sn = sp.block_diag([sp.identity(3 * nvc, format="csr"),
                                sp.bsr_matrix((blocks, np.arange(nfc), np.arange(nfc + 1)), shape=(3 * nfc, 3 * nfc))],
                               format="csr")
            nvf, nff = mf.num_vertices, mf.num_faces
            vnf, fnf = Vf.vertex_nodes.astype(np.int64), Vf.face_nodes.astype(np.int64)
            crow = np.concatenate([vnf, fnf, np.repeat(fnf, 3)])
            ccol = np.concatenate([np.arange(nvf), nvf + np.arange(nff), mf.faces.ravel()])
            combine_f = sp.csr_matrix((np.concatenate([np.ones(nvf + nff), np.full(3 * nff, 1.0 / 3.0)]), (crow, ccol)),
                                      shape=(Vf.num_nodes, nvf + nff))
            self._bub = (sp.kron(split_c, I3, format="csr"), Sn, combine_f)
        split3, Sn, combine_f = self._bub
        C = combine_f[np.asarray(nodes, dtype=np.int64)]                  # (len(nodes), nvf + nff)
        used = np.unique(C.indices)                                       # hierarchical fine dofs these rows draw from
        nvf = mf.num_vertices
        uv, uf = used[used < nvf], used[used >= nvf] - nvf
        p1 = NodalElement(3, 1, False)
        P_p1 = self._nodal_rows(uv, mf.cells, mc.cells, p1, p1, mf.num_vertices, mc.num_vertices, key="p1")

        class _FB(object):
            node_bary = Vc.element.node_bary[-4:]

            @staticmethod
            def tabulate(lam):
                return NodalElement._bubbles(np.atleast_2d(lam))

        P_fb = self._nodal_rows(uf, mf.cell_faces, mc.cell_faces, _FB, _FB, mf.num_faces, mc.num_faces, key="fb")
        P_h = sp.block_diag([P_p1, P_fb], format="csr")                   # (len(used), nvc + nfc)
        Cu = C[:, used]
        P = sp.kron(Cu, I3, format="csr") @ sp.kron(P_h, I3, format="csr") @ Sn @ split3
        P = P.tocsr()
        P.data[np.abs(P.data) < 1e-14] = 0.0
        P.eliminate_zeros()
        P.sort_indices()
        return P

    def select_rows(self, nodes):
        from .problem import BSR
        d = self.bs
        if self.bubble:
            P = self._bubble_rows(nodes)
            B = sp.bsr_matrix(P, blocksize=(d, d))
            B.sort_indices()
            return BSR(len(nodes), self.nbcols, d, B.indptr, B.indices, B.data)
        P = self._nodal_rows(nodes)
        return BSR(len(nodes), self.nbcols, d, P.indptr, P.indices, P.data[:, None, None] * np.eye(d)[None])

    def cols_of_row_range(self, lo, hi):
        if hi <= lo:
            return np.zeros(0, dtype=np.int64)
        if not self.bubble:      # the scalar rows carry the sparsity: no need to expand their d x d blocks for it
            return self._nodal_rows(np.arange(lo, hi)).indices
        return self.select_rows(np.arange(lo, hi)).colidx

    def transpose(self):
        return LazyTransposed(self)


class LazyTransposed(object):
    is_lazy = True

    def __init__(self, base):
        self.base = base
        self.bs, self.nbrows, self.nbcols = base.bs, base.nbcols, base.nbrows

    def transpose(self):
        return self.base


class LazyTransfer(object):
    """The pieces of ``problem.TransferData`` alfi_amd.dist works with, for one level pair, values on demand."""

    is_lazy = True

    def __init__(self, Vc, Vf, nu, gamma, graph, geometry, tensors):
        from .fespace import coarse_cell_blocks
        d = Vf.dim
        self.Vc, self.Vf = Vc, Vf
        self.blk_nodes = coarse_cell_blocks(Vf)
        self.blk_dofs = np.ascontiguousarray(Vf.node_dofs(self.blk_nodes), dtype=np.int32)
        self.D_I = LazyInteriorRows(Vf, graph[0], graph[1], geometry, tensors, self.blk_nodes)
        el = Vf.element
        bubble = d == 3 and el.bubble and el.degree == 1
        self.P = LazyProlongation(Vc, Vf, bubble=bubble)
        self.PT = self.P.transpose()
        self.PT_plain = LazyProlongation(Vc, Vf, bubble=False).transpose() if bubble else self.PT
        self.nu, self.gamma = nu, gamma
        self.n_f, self.n_c = Vf.num_dofs, Vc.num_dofs
        self.bc_dofs_f, self.bc_dofs_c = Vf.bc_dofs, Vc.bc_dofs
        self._geom, self._tens = geometry, tensors
        self._inject_map = None

    @property
    def inject_map(self):
        """fine node coinciding with every coarse node (firedrake.inject on the nested hierarchy, alfi/solver.py:595)"""
        if self._inject_map is None:
            from .fespace import injection_map
            self._inject_map = injection_map(self.Vc, self.Vf)
        return self._inject_map

    def interior_mats(self, blocks):
        """(K_II, D_II) of the coarse-cell blocks ``blocks``: (len(blocks), m, m) each (forms of alfi/transfer.py:319-324)."""
        Vf = self.Vf
        blocks = np.asarray(blocks, dtype=np.int64)
        nch = 2 ** Vf.dim
        cells = (blocks[:, None] * nch + np.arange(nch)).ravel()
        g, vol = self._geom
        return _hostlib.interior_blocks(Vf.cell_nodes[cells], g[cells], vol[cells], self._tens, Vf.dim,
                                        self.blk_nodes[blocks], Vf.num_nodes, nch)
