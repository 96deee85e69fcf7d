"""Synthetic lid-driven-cavity workloads: everything the multigrid hot path consumes, generated without Firedrake.

Mirrors the roles of alfi/problem.py:5-58 (``NavierStokesProblem`` hooks), examples/ldc2d/ldc2d.py:7-39 and
examples/ldc3d/ldc3d.py:7-31 (geometry, boundary conditions, lid profile) and of the Firedrake machinery that, in the
reference, rediscretises the velocity block on every level (alfi/solver.py:565-568 linearised about the current
state; coarse levels via dmhooks + ``inject`` of the state, solver.py:595).  The "current state" here is the nodal
interpolant of the regularised lid profile extended into the cavity -- deterministic, no RNG.

Output per level: block-CSR operator with Dirichlet rows/columns replaced by the identity, Dirichlet dof list,
vertex-star patches; per level pair: prolongation matrices, grad-div matrix rows of the coarse-cell interior dofs
and the dense interior blocks the Schoeberl transfer inverts.
"""
import time
import numpy as np
import scipy.sparse as sp

from . import _hostlib
from .elements import velocity_element
from .fespace import (VectorFunctionSpace, vector_prolongation, nodal_prolongation, coarse_cell_blocks,
                      injection_map)
from .mesh import rectangle_mesh, box_mesh, mesh_hierarchy


class NavierStokesProblem(object):
    """Hook names follow alfi/problem.py:5-58."""

    def mesh(self, distribution_parameters=None):
        raise NotImplementedError

    def mesh_hierarchy(self, hierarchy, nref, callbacks=None, distribution_parameters=None):
        if hierarchy != "uniform":
            raise NotImplementedError("only the uniform hierarchy is built (bary: SURVEY.md section 8(f))")
        return mesh_hierarchy(self.mesh(distribution_parameters), nref)

    def driver(self, x):
        raise NotImplementedError

    def has_nullspace(self):
        return True

    def char_length(self):
        return 1.0

    def char_velocity(self):
        return 1.0

    def relaxation_direction(self):
        return None

    def dirichlet_facets(self, centroids):
        """Which boundary facets carry the velocity Dirichlet condition (``bcs``, problem.py:27): all of them unless a
        problem overrides this."""
        return np.ones(centroids.shape[0], dtype=bool)


class TwoDimLidDrivenCavityProblem(NavierStokesProblem):
    """examples/ldc2d/ldc2d.py:7-39."""

    def __init__(self, baseN, diagonal=None, regularised=True):
        self.baseN = baseN
        self.diagonal = diagonal or "left"
        self.regularised = regularised
        self.dim = 2

    def mesh(self, distribution_parameters=None):
        return rectangle_mesh(self.baseN, self.baseN, 2.0, 2.0, self.diagonal)

    def driver(self, x):
        w = np.zeros_like(x)
        if self.regularised:
            w[:, 0] = x[:, 0] ** 2 * (2 - x[:, 0]) ** 2 * (0.25 * x[:, 1] ** 2)
        else:
            w[:, 0] = 0.25 * x[:, 1] ** 2
        return w

    def char_length(self):
        return 2.0

    def relaxation_direction(self):
        return "0+:1-"


class ThreeDimLidDrivenCavityProblem(NavierStokesProblem):
    """examples/ldc3d/ldc3d.py:7-31."""

    def __init__(self, baseN):
        self.baseN = baseN
        self.dim = 3

    def mesh(self, distribution_parameters=None):
        return box_mesh(self.baseN, self.baseN, self.baseN, 2.0, 2.0, 2.0)

    def driver(self, x):
        w = np.zeros_like(x)
        w[:, 0] = (x[:, 0] ** 2 * (2 - x[:, 0]) ** 2 * x[:, 2] ** 2 * (2 - x[:, 2]) ** 2 * (0.25 * x[:, 1] ** 2))
        return w

    def char_length(self):
        return 2.0

    def relaxation_direction(self):
        return "0+:1-"


class ThreeDimBackwardsFacingStepProblem(NavierStokesProblem):
    """examples/bfs3d/bfs3d.py:8-33 on the structured stand-in for the gmsh channel (``mesh.bfs3d_mesh``): Poiseuille
    inflow at x = 0 (label 1), no-slip walls (label 3), natural outflow at x = 10 (label 2)."""

    def __init__(self, baseN=1, msh=None):
        """msh: path of a gmsh 2.2 ASCII mesh of the channel (the reference's ``--mesh coarse30.msh``, bfs3d.py:13-16,
        37-39); None: the structured stand-in with baseN cubes per unit length."""
        self.baseN = baseN
        self.msh = msh
        self.dim = 3

    def mesh(self, distribution_parameters=None):
        from .mesh import bfs3d_mesh, read_gmsh
        if not self.msh:
            return bfs3d_mesh(self.baseN)
        mesh = read_gmsh(self.msh)
        self.check_boundary_tags(mesh)
        return mesh

    def check_boundary_tags(self, mesh):
        """The reference applies its boundary conditions by physical tag (1 = inflow, 3 = walls Dirichlet, 2 = outflow natural,
        bfs3d.py:23-26); here the Dirichlet part of the boundary is chosen geometrically (``dirichlet_facets``: everything
        but the outflow plane x = 10), because the refined levels of the hierarchy carry no tags.  A gmsh channel of another
        length or orientation would silently get wrong conditions: refuse a mesh whose tags disagree with the predicate."""
        tags = getattr(mesh, "boundary_tags", None)
        if not tags:
            return
        bf = mesh.boundary_facets
        keys = [tuple(sorted(int(v) for v in mesh.facets[f])) for f in bf]
        tagged = np.array([tags.get(kk, -1) for kk in keys])
        geo = np.asarray(self.dirichlet_facets(mesh.coords[mesh.facets[bf]].mean(axis=1)), dtype=bool)
        known = tagged >= 0
        by_tag = np.isin(tagged, (1, 3))
        bad = known & (by_tag != geo)
        if bad.any():
            raise ValueError("%s: %d boundary facets whose physical tag (1 / 3 = Dirichlet, 2 = natural outflow, bfs3d.py:23-26) "
                             "disagrees with the geometric choice of the Dirichlet boundary (x < 10): this is not the "
                             "reference's channel" % (self.msh, int(bad.sum())))

    def driver(self, x):
        """poiseuille_flow (bfs3d.py:19-21): the inflow profile, extended along the channel as the linearisation state."""
        w = np.zeros_like(x)
        y, z = x[:, 1], x[:, 2]
        w[:, 0] = 16.0 * (2 - y) * (y - 1) * z * (1 - z) * (y > 1)
        return w

    def dirichlet_facets(self, centroids):
        return centroids[:, 0] < 10.0 - 1e-9

    def has_nullspace(self):
        return False

    def relaxation_direction(self):
        return "0+:1-"


class BSR(object):
    """Plain block-CSR container (node rows x node cols, bs x bs row-major blocks)."""

    def __init__(self, nbrows, nbcols, bs, rowptr, colidx, vals, bs_col=None):
        self.nbrows, self.nbcols, self.bs = int(nbrows), int(nbcols), int(bs)
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        self.colidx = np.ascontiguousarray(colidx, dtype=np.int32)
        # vals None: the sparsity alone (operator values formed on the device, build_hierarchy(operator_values=False))
        self.vals = None if vals is None else np.ascontiguousarray(vals, dtype=np.float64).reshape(-1, bs, bs)

    @property
    def shape(self):
        return (self.nbrows * self.bs, self.nbcols * self.bs)

    @property
    def nnzb(self):
        return self.colidx.shape[0]

    def to_scipy(self):
        return sp.bsr_matrix((self.vals, self.colidx, self.rowptr), shape=self.shape)

    @staticmethod
    def from_scipy(M, bs):
        B = sp.bsr_matrix(M, blocksize=(bs, bs))
        B.sort_indices()
        return BSR(B.shape[0] // bs, B.shape[1] // bs, bs, B.indptr, B.indices, B.data)

    def select_rows(self, rows):
        """Block rows ``rows`` (in that order) as a new BSR."""
        rows = np.asarray(rows, dtype=np.int64)
        cnt = (self.rowptr[rows + 1] - self.rowptr[rows]).astype(np.int64)
        ptr = np.concatenate([[0], np.cumsum(cnt)])
        idx = np.repeat(self.rowptr[rows].astype(np.int64) - ptr[:-1], cnt) + np.arange(ptr[-1])
        return BSR(len(rows), self.nbcols, self.bs, ptr, self.colidx[idx], None if self.vals is None else self.vals[idx])

    def row_range(self, lo, hi):
        """Block rows lo .. hi-1 as a BSR over the same arrays (views, no copy)."""
        k0, k1 = int(self.rowptr[lo]), int(self.rowptr[hi])
        return BSR(hi - lo, self.nbcols, self.bs, self.rowptr[lo:hi + 1] - self.rowptr[lo], self.colidx[k0:k1], self.vals[k0:k1])

    def transpose(self):
        rp, ci, v = _hostlib.bsr_transpose(self.nbrows, self.nbcols, self.bs, self.rowptr, self.colidx, self.vals)
        return BSR(self.nbcols, self.nbrows, self.bs, rp, ci, v)


class LevelData(object):
    pass


class TransferData(object):
    pass


def build_transfer_data(Vc, Vf, nu, gamma, graph=None):
    """Everything the Schoeberl transfer between two nested spaces needs (alfi/transfer.py:194-259): standard
    prolongation P (bubble-corrected for 3-D P1+FB, transfer.py:334-356) and its transpose, the plain nodal P^T for
    non-robust restriction (solver.py:595), coarse-cell interior blocks (transfer.py:13-46), their dense K / D blocks
    (forms of transfer.py:319-332) and the interior rows of the grad-div matrix."""
    mesh, d, dim = Vf.mesh, Vf.dim, Vf.dim
    element = Vf.element
    rowptr, colidx = graph if graph is not None else _hostlib.node_graph(Vf.cell_nodes, Vf.num_nodes)
    g, vol = mesh.cell_geometry()
    tens = element.reference_tensors()
    T = TransferData()
    blk_nodes = coarse_cell_blocks(Vf)                      # (coarse cells, interior nodes), parent-major
    T.blk_dofs = np.ascontiguousarray(Vf.node_dofs(blk_nodes), dtype=np.int32)     # (nblk, m)
    # dense interior blocks of (2 sym grad u, grad v) and (cell_avg div u, div v), assembled directly
    T.K_II, T.D_II = _hostlib.interior_blocks(Vf.cell_nodes, g, vol, tens, d, blk_nodes, Vf.num_nodes, 2 ** dim)
    # rows of the grad-div matrix (gamma = 1, no BCs) for the interior nodes, in block order
    rows = blk_nodes.ravel().astype(np.int64)
    cnt = (rowptr[rows + 1] - rowptr[rows]).astype(np.int64)
    ptr = np.concatenate([[0], np.cumsum(cnt)])
    idx = np.repeat(rowptr[rows].astype(np.int64) - ptr[:-1], cnt) + np.arange(ptr[-1])
    di_rowptr, di_colidx = ptr.astype(np.int32), colidx[idx]
    del idx
    row_map = np.full(Vf.num_nodes, -1, dtype=np.int32)
    row_map[rows] = np.arange(rows.shape[0], dtype=np.int32)
    di_vals = _hostlib.assemble_bsr(Vf.cell_nodes, g, vol, tens, d, di_rowptr, di_colidx, gamma=1.0, row_map=row_map)
    T.D_I = BSR(rows.shape[0], Vf.num_nodes, d, di_rowptr, di_colidx, di_vals)
    T.D_IT = T.D_I.transpose()                            # (fine nodes) x (interior nodes)
    def scalar_blocks(Ps):       # kron(Ps, I_d) as BSR without forming the d^2-times larger scalar matrix
        return BSR(Ps.shape[0], Ps.shape[1], d, Ps.indptr, Ps.indices, Ps.data[:, None, None] * np.eye(d)[None])

    if dim == 3 and element.bubble and element.degree == 1:
        T.P = BSR.from_scipy(vector_prolongation(Vc, Vf), d)
        T.PT = T.P.transpose()
        T.PT_plain = scalar_blocks(nodal_prolongation(Vc, Vf)).transpose()
    else:
        T.P = scalar_blocks(nodal_prolongation(Vc, Vf))
        T.PT = T.P.transpose()
        T.PT_plain = T.PT
    T.inject_map = injection_map(Vc, Vf)
    T.nu, T.gamma = nu, gamma
    T.n_f, T.n_c = Vf.num_dofs, Vc.num_dofs
    T.bc_dofs_f, T.bc_dofs_c = Vf.bc_dofs, Vc.bc_dofs
    return T


def build_pressure_coupling(L, zero_bc_columns=True, both=False):
    """P0 pressure space of the finest level (solver.py:574-586: ``Q = FunctionSpace(mesh, "DG", 0)``): the discrete
    divergence B (cells x velocity dofs, B[c, (a, x)] = -int_c d_x phi_a, Dirichlet velocity columns zeroed) and the
    diagonal pressure mass matrix (cell volumes).  With these the augmented-Lagrangian term of the level operator is
    gamma B^T M_p^-1 B (solver.py:565-568: ``gamma * inner(cell_avg(div(u)), div(v))``).  Returns scipy CSR B, vol; with
    ``both``: (B with the Dirichlet columns zeroed, B with all columns, vol)."""
    V = L.V
    mesh, d, el = V.mesh, V.dim, V.element
    g, vol = mesh.cell_geometry()
    bI = el.reference_tensors()["bI"]                          # (nloc, d+1): cell average of d_i phi_a
    bdiv = np.einsum("cix,ai->cax", g, bI)                     # cell average of d_x phi_a
    nc, nloc = V.cell_nodes.shape
    # every row holds the nloc * d dofs of one cell: the CSR arrays directly (no COO pass, no duplicates)
    cols = (V.cell_nodes[:, :, None] * d + np.arange(d, dtype=V.cell_nodes.dtype)[None, None, :]).reshape(nc, -1)
    vals = (-vol[:, None, None] * bdiv).reshape(nc, -1)
    itype = np.int32 if nc * nloc * d < 2 ** 31 else np.int64

    nonzero = vals != 0.0

    def csr(keep):             # the entries with keep (None: all) and a non-zero value
        nz = nonzero if keep is None else nonzero & keep[cols]
        ptr = np.concatenate([[0], np.cumsum(nz.sum(axis=1))])
        M = sp.csr_matrix((vals[nz], cols[nz].astype(itype), ptr.astype(itype)), shape=(nc, V.num_dofs))
        M.sort_indices()
        return M

    def free_columns():        # the Jacobian's block; with all columns it is the divergence used in the nonlinear residual
        keep = np.ones(V.num_dofs, dtype=bool)
        keep[V.bc_dofs] = False
        return keep
    if both:                   # (B with the Dirichlet columns zeroed, B with all columns, vol): one pass over the cells
        return csr(free_columns()), csr(None), vol
    return csr(free_columns() if zero_bc_columns else None), vol


def build_hierarchy(problem, nref, k, Re, gamma=1e4, advect=True, patches=True, verbose=False, lazy=False,
                    operator_values=True):
    """Levels 0..nref of the velocity block for ``problem`` at Reynolds number Re.

    nu = char_length * char_velocity / Re (alfi/solver.py:261-267); gamma default 1e4 (alfi/driver.py:30).

    lazy: rank-local generation (alfi_amd.lazy): meshes, numbering, graphs, patches and coarse-cell blocks as usual --
    what the mesh partitioner needs -- but operators and transfers as recipes that assemble the rows a rank asks for
    (``alfi_amd.dist.DistMultigrid`` then never holds global values).

    operator_values False: the level operators as sparsity only (``L.A.vals`` None) -- for a caller that forms them on the
    device (alfi_level_assemble; HipNavierStokesSolver does, every Newton step) and has no use for a host copy."""
    t0 = time.time()
    dim = problem.dim
    element = velocity_element(dim, k)
    mh = problem.mesh_hierarchy("uniform", nref)
    nu = problem.char_length() * problem.char_velocity() / Re if Re > 0 else problem.char_length() * problem.char_velocity()
    adv = 1.0 if (advect and Re > 0) else 0.0
    levels, transfers = [], []
    Vprev = None
    for l, mesh in enumerate(mh):
        V = VectorFunctionSpace(mesh, element, dirichlet=getattr(problem, "dirichlet_facets", None))
        d = V.dim
        L = LevelData()
        L.V, L.level, L.n, L.bs = V, l, V.num_dofs, d
        rowptr, colidx = _hostlib.node_graph(V.cell_nodes, V.num_nodes)
        g, vol = mesh.cell_geometry()
        tens = element.reference_tensors()
        bcmask = np.repeat(V.bc_node_mask, d)
        wind = problem.driver(V.node_coords)
        if lazy:
            from .lazy import LazyOperator, LazyTransfer
            L.A = LazyOperator(V, rowptr, colidx, (g, vol), tens, nu, gamma, adv, wind, values=operator_values)
        elif not operator_values:
            L.A = BSR(V.num_nodes, V.num_nodes, d, rowptr, colidx, None)
        else:
            # level operator in one pass over the cells
            A = _hostlib.assemble_bsr(V.cell_nodes, g, vol, tens, d, rowptr, colidx, nu=nu, gamma=gamma, adv=adv,
                                      wind=wind if adv else None)
            _hostlib.apply_bc_bsr(V.num_nodes, d, rowptr, colidx, A, bcmask)
            L.A = BSR(V.num_nodes, V.num_nodes, d, rowptr, colidx, A)
        L.bc_dofs = V.bc_dofs
        L.nu, L.gamma = nu, gamma
        if patches and l > 0:
            L.patch_ptr, L.patch_dofs, L.patch_seeds = V.star_patches()
        if l > 0 and lazy:
            transfers.append(LazyTransfer(Vprev, V, nu, gamma, (rowptr, colidx), (g, vol), tens))
        elif l > 0:
            transfers.append(build_transfer_data(Vprev, V, nu, gamma, graph=(rowptr, colidx)))
        levels.append(L)
        Vprev = V
        if verbose:
            print("[alfi_amd] level %d: %d cells, %d dofs, %d block nnz (%.1fs)"
                  % (l, mesh.num_cells, V.num_dofs, L.A.nnzb, time.time() - t0), flush=True)
    return levels, transfers
