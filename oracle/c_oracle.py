"""TEST INFRASTRUCTURE ONLY -- ctypes driver of oracle/libalfi_oracle.so (C/OpenMP restatement, oracle/alfi_oracle.c).

Used by tests (second checker next to the NumPy oracle) and by bench.py's cpu_baseline leg.  Never imported by alfi_amd/.
The cycle structure (PCMG V-cycle, alfi/solver.py:359-379) is composed here in Python from the C kernels; the coarse
solve is scipy's SuperLU (the reference uses SuperLU_DIST, solver.py:369-378)."""
import ctypes
import os

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
vp = ctypes.c_void_p


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libalfi_oracle.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _lib = ctypes.CDLL(path)
        _lib.oracle_num_threads.restype = ctypes.c_int
        _lib.oracle_invert_patches.restype = ctypes.c_int
        _lib.oracle_block_invert.restype = ctypes.c_int
        _lib.oracle_set_num_threads.restype = ctypes.c_int
        # one thread per CPU the process may really use (cgroup share), OMP_NUM_THREADS overrides
        if "OMP_NUM_THREADS" not in os.environ:
            from alfi_amd._hostlib import cpu_share
            _lib.oracle_set_num_threads(ctypes.c_int(cpu_share()))
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(vp)


class _CLevel(ctypes.Structure):
    _fields_ = [("nbrows", ctypes.c_int64), ("bs", ctypes.c_int), ("rowptr", vp), ("colidx", vp), ("vals", vp),
                ("npatch", ctypes.c_int64), ("patch_ptr", vp), ("patch_dofs", vp), ("inv_ptr", vp), ("inv", vp),
                ("dof_ptr", vp), ("dof_pos", vp), ("bc_dofs", vp), ("nbc", ctypes.c_int64), ("stage", vp)]


def spmv(B, x, y=None, b=None, alpha=1.0):
    """y = B x, or y = b - alpha B x when b is given.  B: alfi_amd.problem.BSR."""
    if y is None:
        y = np.empty(B.nbrows * B.bs)
    lib().oracle_bsr_spmv(ctypes.c_int64(B.nbrows), ctypes.c_int(B.bs), _p(B.rowptr), _p(B.colidx), _p(B.vals), _p(x),
                          _p(y), _p(b), ctypes.c_double(alpha), ctypes.c_int(0 if b is None else 1))
    return y


def _numa_copy(a):
    """Copy of a NumPy array whose pages are first touched by the OpenMP threads (static schedule), see
    oracle_parallel_copy in alfi_oracle.c."""
    a = np.ascontiguousarray(a)
    if os.environ.get("ALFI_ORACLE_NUMA", "1") == "0":
        return a
    out = np.empty_like(a)
    if a.nbytes:
        lib().oracle_parallel_copy(_p(out), _p(a), ctypes.c_int64(a.nbytes))
    return out


class _BSRView(object):
    pass


class CLevel(object):
    def __init__(self, L, with_patches=True):
        self.L = L
        # operator arrays re-homed for the timed baseline (NUMA first touch by the threads that will read them)
        self.A = _BSRView()
        self.A.nbrows, self.A.nbcols, self.A.bs = L.A.nbrows, L.A.nbcols, L.A.bs
        self.A.rowptr, self.A.colidx, self.A.vals = L.A.rowptr, _numa_copy(L.A.colidx), _numa_copy(L.A.vals)
        self.n, self.bs = L.n, L.bs
        self.bc = np.ascontiguousarray(L.bc_dofs, dtype=np.int32)
        self.has_patches = with_patches
        if with_patches:
            self.pptr = np.ascontiguousarray(L.patch_ptr, dtype=np.int64)
            self.pdofs = np.ascontiguousarray(L.patch_dofs, dtype=np.int32)
            npt = np.diff(self.pptr)
            self.inv_ptr = np.concatenate([[0], np.cumsum(npt * npt)]).astype(np.int64)
            self.inv = np.empty(self.inv_ptr[-1])
            order = np.argsort(self.pdofs, kind="stable")
            self.dof_ptr = np.concatenate([[0], np.cumsum(np.bincount(self.pdofs, minlength=self.n))]).astype(np.int32)
            self.dof_pos = order.astype(np.int32)
            self.stage = np.empty(self.pptr[-1])
            self.factor()
            self.c = _CLevel(self.A.nbrows, self.bs, _p(self.A.rowptr), _p(self.A.colidx), _p(self.A.vals),
                             len(npt), _p(self.pptr), _p(self.pdofs), _p(self.inv_ptr), _p(self.inv),
                             _p(self.dof_ptr), _p(self.dof_pos), _p(self.bc), len(self.bc), _p(self.stage))
        self.work = None

    def factor(self):
        rc = lib().oracle_invert_patches(ctypes.c_int64(len(self.pptr) - 1), _p(self.pptr), _p(self.pdofs),
                                         ctypes.c_int(self.bs), _p(self.A.rowptr), _p(self.A.colidx), _p(self.A.vals),
                                         _p(self.inv_ptr), _p(self.inv))
        if rc != 0:
            raise RuntimeError("singular patch")

    def patch_apply(self, x):
        y = np.empty(self.n)
        lib().oracle_patch_apply(ctypes.c_int64(len(self.pptr) - 1), _p(self.pptr), _p(self.pdofs), _p(self.inv_ptr),
                                 _p(self.inv), ctypes.c_int64(self.n), _p(self.dof_ptr), _p(self.dof_pos), _p(self.bc),
                                 ctypes.c_int64(len(self.bc)), _p(x), _p(y), _p(self.stage))
        return y

    def smooth(self, k, b, x, nonzero_guess=True):
        if self.work is None or self.work.shape[0] < (2 * k + 2) * self.n:
            self.work = np.empty((2 * k + 2) * self.n)
        lib().oracle_fgmres(ctypes.byref(self.c), ctypes.c_int(k), _p(b), _p(x), ctypes.c_int(1 if nonzero_guess else 0),
                            _p(self.work))
        return x


class CTransfer(object):
    def __init__(self, T):
        self.T = T
        self.nblk, self.m = T.blk_dofs.shape
        self.blk = np.ascontiguousarray(T.blk_dofs, dtype=np.int32)
        self.binv = np.empty((self.nblk, self.m, self.m))
        rc = lib().oracle_block_invert(ctypes.c_int64(self.nblk), ctypes.c_int(self.m),
                                       _p(np.ascontiguousarray(T.K_II)), _p(np.ascontiguousarray(T.D_II)),
                                       ctypes.c_double(T.nu), ctypes.c_double(T.gamma), _p(self.binv))
        if rc != 0:
            raise RuntimeError("singular transfer block")

    def _block_gemv(self, vec, gather):
        out = np.empty(self.nblk * self.m)
        lib().oracle_block_gemv(ctypes.c_int64(self.nblk), ctypes.c_int(self.m), _p(self.binv), _p(self.blk), _p(vec),
                                _p(out), ctypes.c_int(1 if gather else 0))
        return out

    def prolong(self, xc):
        T = self.T
        xf = spmv(T.P, xc)
        t = self._block_gemv(spmv(T.D_I, xf), False)
        xf[self.blk.ravel()] -= T.gamma * t
        xf[T.bc_dofs_f] = 0.0
        return xf

    def restrict(self, rf, robust):
        T = self.T
        if robust:
            t = self._block_gemv(rf, True)
            s = spmv(T.D_IT, t, b=rf, alpha=T.gamma)
            rc = spmv(T.PT, s)
        else:
            rc = spmv(T.PT_plain, rf)
        rc[T.bc_dofs_c] = 0.0
        return rc


class CMultigrid(object):
    def __init__(self, levels, transfers, k, robust_restriction=False):
        self.k, self.robust = k, robust_restriction
        self.levels = [CLevel(L, with_patches=L.level > 0) for L in levels]
        self.transfers = [CTransfer(T) for T in transfers]
        self.coarse = spla.splu(sp.csc_matrix(levels[0].A.to_scipy()))

    def vcycle(self, l, b, x):
        if l == 0:
            return self.coarse.solve(b)
        L, T = self.levels[l], self.transfers[l - 1]
        x = L.smooth(self.k, b, x)
        r = spmv(L.A, x, b=b, alpha=1.0)
        bc = T.restrict(r, self.robust)
        xc = self.vcycle(l - 1, bc, np.zeros_like(bc))
        x += T.prolong(xc)
        return L.smooth(self.k, b, x)
