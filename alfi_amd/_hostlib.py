"""ctypes binding of libalfi_host.so (CPU operator generator, csrc/host_assemble.cpp).  Input generation only."""
import ctypes
import os
import numpy as np

_lib = None


def cpu_share():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box hands out a share
    of its cores; OpenMP teams larger than the quota are throttled into the ground)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = float(txt[0])
                per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libalfi_host.so")
        if not os.path.exists(path):
            from . import build
            build.build_host()
        _lib = ctypes.CDLL(path)
        _lib.alfi_host_node_graph.restype = ctypes.c_int
        _lib.alfi_host_assemble_bsr.restype = ctypes.c_int
        _lib.alfi_host_apply_bc_bsr.restype = ctypes.c_int
        _lib.alfi_host_extract_blocks.restype = ctypes.c_int
        _lib.alfi_host_interior_blocks.restype = ctypes.c_int
        _lib.alfi_host_bsr_transpose.restype = ctypes.c_int
        _lib.alfi_host_supg.restype = ctypes.c_int
        # ALFI_HOST_THREADS overrides OMP_NUM_THREADS (torch.distributed.run exports OMP_NUM_THREADS=1 to every rank)
        nthr = int(os.environ.get("ALFI_HOST_THREADS", "0")) or cpu_share()
        _lib.alfi_host_set_num_threads(ctypes.c_int(nthr))
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def node_graph(cell_nodes, nnode):
    cn = np.ascontiguousarray(cell_nodes, dtype=np.int32)
    ncell, nloc = cn.shape
    rowptr = np.zeros(nnode + 1, dtype=np.int32)
    rc = lib().alfi_host_node_graph(ctypes.c_int64(ncell), ctypes.c_int(nloc), _p(cn), ctypes.c_int64(nnode),
                                    _p(rowptr), None)
    if rc != 0:
        raise RuntimeError("node graph exceeds int32 indexing")
    colidx = np.empty(rowptr[-1], dtype=np.int32)
    lib().alfi_host_node_graph(ctypes.c_int64(ncell), ctypes.c_int(nloc), _p(cn), ctypes.c_int64(nnode), _p(rowptr),
                               _p(colidx))
    return rowptr, colidx


def assemble_bsr(cell_nodes, g, vol, tensors, d, rowptr, colidx, nu=0.0, gamma=0.0, adv=0.0, wind=None, out=None,
                 row_map=None, gamma_full=0.0):
    """gamma: coefficient of (cell_avg div u, div v) (PkP0 forms); gamma_full: of (div u, div v) (Scott-Vogelius forms)."""
    cn = np.ascontiguousarray(cell_nodes, dtype=np.int32)
    ncell, nloc = cn.shape
    g = np.ascontiguousarray(g, dtype=np.float64)
    vol = np.ascontiguousarray(vol, dtype=np.float64)
    S, bI, T1 = (np.ascontiguousarray(tensors[k], dtype=np.float64) for k in ("S", "bI", "T1"))
    if wind is not None:
        wind = np.ascontiguousarray(wind, dtype=np.float64)
    if out is None:
        out = np.zeros((colidx.shape[0], d, d), dtype=np.float64)
    if row_map is not None:
        row_map = np.ascontiguousarray(row_map, dtype=np.int32)
    rc = lib().alfi_host_assemble_bsr(ctypes.c_int64(ncell), ctypes.c_int(nloc), ctypes.c_int(d), _p(cn), _p(g),
                                      _p(vol), _p(S), _p(bI), _p(T1), _p(wind), ctypes.c_double(nu),
                                      ctypes.c_double(gamma), ctypes.c_double(adv), _p(row_map), _p(rowptr), _p(colidx),
                                      _p(out), ctypes.c_double(gamma_full))
    if rc != 0:
        raise RuntimeError("assemble_bsr failed (%d): sparsity pattern does not cover the mesh" % rc)
    return out


def cell_size(mesh):
    """Firedrake's ``CellSize`` = 2 * circumradius [3P] (problem.mesh_size(u, "cell"), alfi/problem.py:46-52)."""
    x = mesh.coords[mesh.cells]
    if mesh.dim == 2:
        a = np.linalg.norm(x[:, 1] - x[:, 2], axis=1)
        b = np.linalg.norm(x[:, 0] - x[:, 2], axis=1)
        c = np.linalg.norm(x[:, 0] - x[:, 1], axis=1)
        area = mesh.cell_geometry()[1]
        return 2.0 * a * b * c / (4.0 * area)
    e = lambda i, j: np.linalg.norm(x[:, i] - x[:, j], axis=1)
    aA, bB, cC = e(0, 1) * e(2, 3), e(0, 2) * e(1, 3), e(0, 3) * e(1, 2)         # products of opposite edges
    vol = mesh.cell_geometry()[1]
    rad = np.sqrt((aA + bB + cC) * (aA + bB - cC) * (aA - bB + cC) * (-aA + bB + cC)) / (24.0 * vol)
    return 2.0 * rad


def supg(V, U, nu, weight, magic, rowptr=None, colidx=None, vals=None, F=None, nq=None):
    """SUPG stabilisation (stabilisation.py:47-97, solver.py:204-234) about the state U (num_nodes, dim): adds the residual
    contribution to F (num_dofs) and / or the Newton linearisation to the BSR values ``vals``.  Quadrature degree 2k."""
    from .elements import simplex_quadrature
    mesh, el, d = V.mesh, V.element, V.dim
    deg = 3 if (el.bubble or el.degree == 3) else el.degree                  # ufl degree of the (enriched) element
    lam, wq = simplex_quadrature(d, nq or (deg + 1))                          # n points per direction: exact to 2n - 1 >= 2k
    phi, dphi = el.tabulate(lam)
    d2phi = el.tabulate_hessian(lam)
    g, vol = mesh.cell_geometry()
    h = cell_size(mesh)
    cn = np.ascontiguousarray(V.cell_nodes, dtype=np.int32)
    U = np.ascontiguousarray(U, dtype=np.float64)
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (g, vol, h, wq, phi, dphi, d2phi)]
    rc = lib().alfi_host_supg(ctypes.c_int64(cn.shape[0]), ctypes.c_int(cn.shape[1]), ctypes.c_int(d), _p(cn), _p(arrs[0]),
                              _p(arrs[1]), _p(arrs[2]), ctypes.c_int(len(wq)), _p(arrs[3]), _p(arrs[4]), _p(arrs[5]),
                              _p(arrs[6]), _p(U), ctypes.c_double(nu), ctypes.c_double(weight), ctypes.c_double(magic),
                              _p(rowptr), _p(colidx), _p(vals), _p(F))
    if rc != 0:
        raise RuntimeError("supg failed (%d): sparsity pattern does not cover the mesh" % rc)


def contributors(cell_nodes, nnode, rowptr, colidx, nindex=None, partial=False):
    """(cptr int64 (nnzb + 1), ccell int32, cba uint16): for every BSR block the (cell, b * nloc + a) pairs that contribute to
    it, in a fixed order -- the gather lists of the device assembly (alfi_level_set_assembly).  nindex: cell_nodes may index
    up to nindex > nnode nodes (only the first nnode have operator rows); partial: pairs without a block are skipped (a
    rank's ghost rows hold local columns only)."""
    cn = np.ascontiguousarray(cell_nodes, dtype=np.int32)
    ncell, nloc = cn.shape
    if nloc * nloc > 65535:
        raise ValueError("element with %d nodes: the pair code does not fit 16 bits" % nloc)
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    colidx = np.ascontiguousarray(colidx, dtype=np.int32)
    cptr = np.zeros(colidx.shape[0] + 1, dtype=np.int64)
    fn = lib().alfi_host_contributors
    fn.restype = ctypes.c_int
    tail = (ctypes.c_int64(int(nindex) if nindex is not None else int(nnode)), ctypes.c_int(1 if partial else 0))
    rc = fn(ctypes.c_int64(ncell), ctypes.c_int(nloc), _p(cn), ctypes.c_int64(nnode), _p(rowptr), _p(colidx), _p(cptr), None, None,
            *tail)
    if rc != 0:
        raise RuntimeError("contributors failed (%d): sparsity pattern does not cover the mesh" % rc)
    assert partial or cptr[-1] == ncell * nloc * nloc
    ccell = np.empty(cptr[-1], dtype=np.int32)
    cba = np.empty(cptr[-1], dtype=np.uint16)
    fn(ctypes.c_int64(ncell), ctypes.c_int(nloc), _p(cn), ctypes.c_int64(nnode), _p(rowptr), _p(colidx), _p(cptr), _p(ccell), _p(cba),
       *tail)
    return cptr, ccell, cba


def apply_bc_bsr(nrow, d, rowptr, colidx, vals, bcmask, row_ids=None):
    """Dirichlet rows / columns -> identity.  row_ids: the node of each block row when the rows are a subset."""
    bcmask = np.ascontiguousarray(bcmask, dtype=np.uint8)
    if row_ids is not None:
        row_ids = np.ascontiguousarray(row_ids, dtype=np.int32)
    lib().alfi_host_apply_bc_bsr(ctypes.c_int64(nrow), ctypes.c_int(d), _p(rowptr), _p(colidx), _p(vals), _p(bcmask),
                                 _p(row_ids))


def extract_blocks(d, rowptr, colidx, vals, blk_ptr, blk_dofs):
    """Dense A[dofs_b, dofs_b] for every block; returns (out_ptr, flat out)."""
    blk_ptr = np.ascontiguousarray(blk_ptr, dtype=np.int64)
    blk_dofs = np.ascontiguousarray(blk_dofs, dtype=np.int32)
    n = np.diff(blk_ptr)
    out_ptr = np.concatenate([[0], np.cumsum(n * n)]).astype(np.int64)
    out = np.empty(out_ptr[-1], dtype=np.float64)
    lib().alfi_host_extract_blocks(ctypes.c_int(d), _p(rowptr), _p(colidx), _p(vals), ctypes.c_int64(len(n)),
                                   _p(blk_ptr), _p(blk_dofs), _p(out_ptr), _p(out))
    return out_ptr, out


def interior_blocks(cell_nodes, g, vol, tensors, d, blk_nodes, num_nodes, nch, full_div=False):
    """K_II, D_II (nblk, m, m) of the coarse-cell interior dofs; children of block b are cells b*nch .. b*nch+nch-1."""
    cn = np.ascontiguousarray(cell_nodes, dtype=np.int32)
    nloc = cn.shape[1]
    nblk, mn = blk_nodes.shape
    m = mn * d
    blk_local = np.full(num_nodes, -1, dtype=np.int32)
    blk_local[blk_nodes.ravel()] = np.tile(np.arange(mn, dtype=np.int32), nblk)
    g = np.ascontiguousarray(g, dtype=np.float64)
    vol = np.ascontiguousarray(vol, dtype=np.float64)
    S, bI = (np.ascontiguousarray(tensors[k], dtype=np.float64) for k in ("S", "bI"))
    K = np.empty((nblk, m, m))
    D = np.empty((nblk, m, m))
    lib().alfi_host_interior_blocks(ctypes.c_int64(nblk), ctypes.c_int(nch), ctypes.c_int(nloc), ctypes.c_int(d),
                                    _p(cn), _p(g), _p(vol), _p(S), _p(bI), _p(blk_local), ctypes.c_int(m), _p(K), _p(D),
                                    ctypes.c_int(1 if full_div else 0))
    return K, D


def bsr_transpose(nbrows, nbcols, bs, rowptr, colidx, vals):
    rowptr_t = np.empty(nbcols + 1, dtype=np.int32)
    colidx_t = np.empty(colidx.shape[0], dtype=np.int32)
    vals_t = np.empty_like(vals)
    lib().alfi_host_bsr_transpose(ctypes.c_int64(nbrows), ctypes.c_int64(nbcols), ctypes.c_int(bs), _p(rowptr),
                                  _p(colidx), _p(vals), _p(rowptr_t), _p(colidx_t), _p(vals_t))
    return rowptr_t, colidx_t, vals_t
