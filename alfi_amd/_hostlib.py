"""ctypes binding of libalfi_host.so (CPU operator generator, csrc/host_assemble.cpp).  Input generation only."""
import ctypes
import os
import numpy as np

_lib = None


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


def assemble_bsr(cell_nodes, g, vol, tensors, d, rowptr, colidx, nu=0.0, gamma=0.0, adv=0.0, wind=None, out=None):
    cn = np.ascontiguousarray(cell_nodes, dtype=np.int32)
    ncell, nloc = cn.shape
    g = np.ascontiguousarray(g, dtype=np.float64)
    vol = np.ascontiguousarray(vol, dtype=np.float64)
    S, bI, T1 = (np.ascontiguousarray(tensors[k], dtype=np.float64) for k in ("S", "bI", "T1"))
    if wind is not None:
        wind = np.ascontiguousarray(wind, dtype=np.float64)
    if out is None:
        out = np.zeros((colidx.shape[0], d, d), dtype=np.float64)
    rc = lib().alfi_host_assemble_bsr(ctypes.c_int64(ncell), ctypes.c_int(nloc), ctypes.c_int(d), _p(cn), _p(g),
                                      _p(vol), _p(S), _p(bI), _p(T1), _p(wind), ctypes.c_double(nu),
                                      ctypes.c_double(gamma), ctypes.c_double(adv), _p(rowptr), _p(colidx), _p(out))
    if rc != 0:
        raise RuntimeError("assemble_bsr failed (%d): sparsity pattern does not cover the mesh" % rc)
    return out


def apply_bc_bsr(nnode, d, rowptr, colidx, vals, bcmask):
    bcmask = np.ascontiguousarray(bcmask, dtype=np.uint8)
    lib().alfi_host_apply_bc_bsr(ctypes.c_int64(nnode), ctypes.c_int(d), _p(rowptr), _p(colidx), _p(vals), _p(bcmask))


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
