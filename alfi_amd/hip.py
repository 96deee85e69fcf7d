"""Thin object layer over the C ABI (include/alfi_hip.h): Context, DeviceVec, Level, Transfer, Multigrid.

All arithmetic happens in libalfi_hip.so; this module only moves arrays across the boundary and keeps handles alive.
Errors from the library become ``RuntimeError`` (the reference's plug-ins raise Python exceptions, solver.py:37-38).
"""
import ctypes
import numpy as np

from . import _lib
from ._lib import BsrHost, CsrHost, vp


class AlfiHipError(RuntimeError):
    pass


def _ptr(a):
    return None if a is None else a.ctypes.data_as(vp)


class Context(object):
    def __init__(self, device=0, stream=None):
        self.lib = _lib.load()
        h = vp()
        rc = self.lib.alfi_ctx_create(int(device), vp(stream) if stream else None, ctypes.byref(h))
        if rc != 0:
            raise AlfiHipError("alfi_ctx_create failed (%d): %s -- alfi_amd needs an MI355X; there is no CPU fallback"
                               % (rc, self.lib.alfi_last_error(None).decode()))
        self.h = h
        self.device = device

    def check(self, rc):
        if rc != 0:
            raise AlfiHipError("libalfi_hip error %d: %s" % (rc, self.lib.alfi_last_error(self.h).decode()))

    def sync(self):
        self.check(self.lib.alfi_ctx_sync(self.h))

    def close(self):
        if self.h:
            self.lib.alfi_ctx_destroy(self.h)
            self.h = None

    def set_assembly_scratch(self, max_bytes):
        """Bytes of element blocks the operator refresh may hold at once (alfi_ctx_set_assembly_scratch); beyond, the cells
        are taken in batches."""
        self.check(self.lib.alfi_ctx_set_assembly_scratch(self.h, int(max_bytes)))

    def set_comm(self, callback, dred_ptr, dred_len):
        """callback: a ctypes function pointer of type alfi_amd.dist.CommFn (kept alive by the caller)."""
        self.check(self.lib.alfi_ctx_set_comm(self.h, ctypes.cast(callback, vp), None, vp(int(dred_ptr)), int(dred_len)))

    def comm_init(self, unique_id, rank, nranks):
        """Native transport: the ctx creates its own RCCL communicator from the 128-byte id (collective over the ranks)."""
        buf = ctypes.create_string_buffer(bytes(unique_id), _lib.COMM_ID_BYTES)
        self.check(self.lib.alfi_ctx_comm_init(self.h, ctypes.cast(buf, vp), int(rank), int(nranks)))

    def comm_allow_self(self, on=True):
        """TEST HOOK (alfi_ctx_comm_allow_self): a rank may name itself as a neighbour."""
        self.check(self.lib.alfi_ctx_comm_allow_self(self.h, 1 if on else 0))

    def comm_size(self):
        """(rank, ranks) of the communicator the library itself exchanges over (alfi_ctx_comm_size)."""
        r, n = ctypes.c_int(), ctypes.c_int()
        self.check(self.lib.alfi_ctx_comm_size(self.h, ctypes.byref(r), ctypes.byref(n)))
        return r.value, n.value

    # vectors ----------------------------------------------------------------------------------------------------------
    def vec(self, n_or_array):
        if isinstance(n_or_array, (int, np.integer)):
            v = DeviceVec(self, int(n_or_array))
            v.zero()
            return v
        a = np.ascontiguousarray(n_or_array, dtype=np.float64)
        v = DeviceVec(self, a.shape[0])
        v.set(a)
        return v

    # profiling (PETSc-event style report, driver.py:77-92) ----------------------------------------------------------------
    def transfer_stats(self, reset=False):
        """(bytes host -> device, bytes device -> host) of every copy the library made since the last reset, process-wide
        (alfi_transfer_stats)."""
        a, b = ctypes.c_int64(), ctypes.c_int64()
        self.check(self.lib.alfi_transfer_stats(ctypes.byref(a), ctypes.byref(b), 1 if reset else 0))
        return a.value, b.value

    def axpy(self, y, x, a, n=None, y_off=0, x_off=0):
        """y[y_off : y_off + n] += a x[x_off : x_off + n] on the device (alfi_vec_axpy)."""
        n = min(y.n - y_off, x.n - x_off) if n is None else int(n)
        self.check(self.lib.alfi_vec_axpy(self.h, vp(y.ptr.value + 8 * y_off), vp(x.ptr.value + 8 * x_off), float(a), n))

    def copy(self, y, x, n=None, y_off=0, x_off=0):
        """y[y_off : y_off + n] = x[x_off : x_off + n] on the device (alfi_vec_copy)."""
        n = min(y.n - y_off, x.n - x_off) if n is None else int(n)
        self.check(self.lib.alfi_vec_copy(self.h, vp(y.ptr.value + 8 * y_off), vp(x.ptr.value + 8 * x_off), n))

    def ivec(self, a):
        """A device array of int32 (index lists of alfi_vec_gather)."""
        return IntVec(self, a)

    def gather(self, dst, src, idx, bs=1):
        """dst[i] = src[idx[i]], bs consecutive doubles per index, on the device (alfi_vec_gather; idx: an IntVec)."""
        assert dst.n >= idx.n * bs
        self.check(self.lib.alfi_vec_gather(self.h, dst.ptr, src.ptr, idx.ptr, idx.n, int(bs)))

    def comm_stats(self, reset=False):
        """(halo exchanges, all-reduces, doubles sent by this rank) since the last reset (alfi_ctx_comm_stats)."""
        a, b, c = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        self.check(self.lib.alfi_ctx_comm_stats(self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), 1 if reset else 0))
        return a.value, b.value, c.value

    def set_graph(self, on=True):
        """Replay whole multigrid cycles as hipGraphs (alfi_ctx_set_graph)."""
        self.check(self.lib.alfi_ctx_set_graph(self.h, 1 if on else 0))

    def prof_enable(self, on=True):
        """True / 1: every event class; 2: PATCH_APPLY and COMM only; False / 0: off."""
        self.check(self.lib.alfi_prof_enable(self.h, int(on)))

    def prof_reset(self):
        self.check(self.lib.alfi_prof_reset(self.h))

    def prof_get(self, level_id=-1, petsc_names=False):
        """{event name: (total device ms, launches)} since the last reset, optionally for one level only; ``petsc_names``: keyed
        by the PETSc log events of the reference's report (alfi/driver.py:80) instead of the library's class names."""
        out = {}
        for i, name in enumerate(_lib.EVENTS):
            ms, cnt = ctypes.c_double(), ctypes.c_int64()
            self.check(self.lib.alfi_prof_get_level(self.h, i, int(level_id), ctypes.byref(ms), ctypes.byref(cnt)))
            out[_lib.PETSC_EVENT_NAMES[name] if petsc_names else name] = (ms.value, cnt.value)
        return out


class DeviceVec(object):
    """A level vector resident in HBM (the analogue of a PETSc Vec's local array)."""

    def __init__(self, ctx, n):
        self.ctx, self.n = ctx, int(n)
        p = vp()
        ctx.check(ctx.lib.alfi_malloc(ctx.h, self.n * 8, ctypes.byref(p)))
        self.ptr = p

    def set(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.shape == (self.n,)
        self.ctx.check(self.ctx.lib.alfi_memcpy_h2d(self.ctx.h, self.ptr, _ptr(a), self.n * 8))

    def get(self):
        out = np.empty(self.n, dtype=np.float64)
        self.ctx.check(self.ctx.lib.alfi_memcpy_d2h(self.ctx.h, _ptr(out), self.ptr, self.n * 8))
        return out

    def zero(self):
        self.ctx.check(self.ctx.lib.alfi_memset0(self.ctx.h, self.ptr, self.n * 8))

    def __del__(self):
        try:
            if self.ptr and self.ctx.h:
                self.ctx.lib.alfi_free(self.ctx.h, self.ptr)
        except Exception:
            pass
        self.ptr = None


class IntVec(object):
    """int32 indices resident on the device."""

    def __init__(self, ctx, a):
        a = np.ascontiguousarray(a, dtype=np.int32)
        self.ctx, self.n = ctx, int(a.shape[0])
        p = vp()
        ctx.check(ctx.lib.alfi_malloc(ctx.h, max(self.n, 1) * 4, ctypes.byref(p)))
        self.ptr = p
        if self.n:
            ctx.check(ctx.lib.alfi_memcpy_h2d(ctx.h, p, _ptr(a), self.n * 4))

    def __del__(self):
        try:
            if self.ptr and self.ctx.h:
                self.ctx.lib.alfi_free(self.ctx.h, self.ptr)
        except Exception:
            pass
        self.ptr = None


def _bsr_struct(B, keep):
    rowptr = np.ascontiguousarray(B.rowptr, dtype=np.int32)
    colidx = np.ascontiguousarray(B.colidx, dtype=np.int32)
    vals = np.ascontiguousarray(B.vals, dtype=np.float64)
    keep.extend([rowptr, colidx, vals])
    return BsrHost(B.nbrows, B.nbcols, _ptr(rowptr), _ptr(colidx), _ptr(vals))


class Level(object):
    """One multigrid level: BSR operator + Dirichlet dofs (+ patches)."""

    def __init__(self, ctx, A, bc_dofs):
        self.ctx = ctx
        self.n, self.bs = A.nbrows * A.bs, A.bs
        bc = np.ascontiguousarray(bc_dofs, dtype=np.int32)
        h = vp()
        # (A.vals None: the sparsity only, the values are formed on the device -- set_assembly + assemble)
        vals = None if A.vals is None else np.ascontiguousarray(A.vals)
        ctx.check(ctx.lib.alfi_level_create(ctx.h, A.nbrows, A.bs, _ptr(A.rowptr), _ptr(A.colidx), _ptr(vals), _ptr(bc), len(bc),
                                            ctypes.byref(h)))
        self.h = h
        self.nnzb = A.nnzb
        i = ctypes.c_int()
        ctx.check(ctx.lib.alfi_level_id(h, ctypes.byref(i)))
        self.id = i.value

    def set_partition(self, nb_owned, distributed, send_nodes, sendbuf_ptr, recvbuf_ptr, nb_ghost):
        """sendbuf_ptr / recvbuf_ptr: device buffers the callback transport exchanges, or None: library-owned."""
        sn = np.ascontiguousarray(send_nodes, dtype=np.int32)
        sp_, rp_ = (vp(int(sendbuf_ptr)), vp(int(recvbuf_ptr))) if sendbuf_ptr is not None else (None, None)
        self.ctx.check(self.ctx.lib.alfi_level_set_partition(self.h, int(nb_owned), 1 if distributed else 0, len(sn),
                                                             _ptr(sn), sp_, rp_, int(nb_ghost)))
        self.n_own = int(nb_owned) * self.bs

    def set_sum_exchange(self, ranks, counts, send_nodes, sum_nodes, sum_ptr, sum_src):
        """Plan of the merged reverse-add + forward exchange of the smoother (alfi_level_set_sum_exchange, native transport)."""
        ranks = np.ascontiguousarray(ranks, dtype=np.int32)
        counts = np.ascontiguousarray(counts, dtype=np.int64)
        send_nodes = np.ascontiguousarray(send_nodes, dtype=np.int32)
        sum_nodes = np.ascontiguousarray(sum_nodes, dtype=np.int32)
        sum_ptr = np.ascontiguousarray(sum_ptr, dtype=np.int32)
        sum_src = np.ascontiguousarray(sum_src, dtype=np.int32)
        assert send_nodes.shape[0] == counts.sum() and sum_ptr.shape[0] == sum_nodes.shape[0] + 1
        self.ctx.check(self.ctx.lib.alfi_level_set_sum_exchange(self.h, len(ranks), _ptr(ranks), _ptr(counts), _ptr(send_nodes),
                                                                len(sum_nodes), _ptr(sum_nodes), _ptr(sum_ptr), _ptr(sum_src)))

    def set_neighbours(self, ranks, send_nodes, recv_nodes):
        """Native transport: neighbour ranks (ascending) and the nodes sent to / received from each."""
        r = np.ascontiguousarray(ranks, dtype=np.int32)
        sc = np.ascontiguousarray(send_nodes, dtype=np.int64)
        rc = np.ascontiguousarray(recv_nodes, dtype=np.int64)
        self.ctx.check(self.ctx.lib.alfi_level_set_neighbours(self.h, len(r), _ptr(r), _ptr(sc), _ptr(rc)))

    def set_overlap(self, nb_interior, npatch_interior):
        self.ctx.check(self.ctx.lib.alfi_level_set_overlap(self.h, int(nb_interior), int(npatch_interior)))

    def update_values(self, vals):
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        self.ctx.check(self.ctx.lib.alfi_level_update_values(self.h, _ptr(vals)))

    def set_assembly(self, V, rowptr, colidx, full_div=False, cells=None, cell_nodes=None):
        """Hand the level what the device-side operator refresh needs (alfi_level_set_assembly): the cells of the nodal
        space ``V`` (alfi_amd.fespace), the reference-cell tensors of its element (viscous / grad-div / advection terms) and
        the contributor lists of the level's sparsity; ``full_div``: the Scott-Vogelius grad-div term (solver.py:616).
        Partitioned level (alfi_amd.dist): ``cells`` = the mesh cells that touch a local node, ``cell_nodes`` their nodes in
        the numbering of the rank's state vector (local nodes first, then the cells' other nodes); rowptr / colidx the LOCAL
        operator's."""
        from . import _hostlib
        g, vol = V.mesh.cell_geometry()
        tens = V.element.reference_tensors()
        S, bI, T1 = (np.ascontiguousarray(tens[k], dtype=np.float64) for k in ("S", "bI", "T1"))
        nrows = len(rowptr) - 1
        if cells is None:
            cn = np.ascontiguousarray(V.cell_nodes, dtype=np.int32)
            cptr, ccell, cba = _hostlib.contributors(cn, V.num_nodes, rowptr, colidx)
        else:
            cn = np.ascontiguousarray(cell_nodes, dtype=np.int32)
            g, vol = g[cells], vol[cells]
            cptr, ccell, cba = _hostlib.contributors(cn, nrows, rowptr, colidx, nindex=int(cn.max()) + 1 if cn.size else nrows,
                                                     partial=True)
        g = np.ascontiguousarray(g, dtype=np.float64)
        vol = np.ascontiguousarray(vol, dtype=np.float64)
        self.ctx.check(self.ctx.lib.alfi_level_set_assembly(self.h, cn.shape[0], cn.shape[1], _ptr(cn), _ptr(g), _ptr(vol),
                                                            _ptr(S), _ptr(bI), _ptr(T1), 1 if full_div else 0, _ptr(cptr),
                                                            _ptr(ccell), _ptr(cba)))

    def set_assembly_bc(self, bc_dofs):
        """Partitioned level: the Dirichlet dofs among ALL local dofs, ghosts included (alfi_level_set_assembly_bc)."""
        b = np.ascontiguousarray(bc_dofs, dtype=np.int32)
        self.ctx.check(self.ctx.lib.alfi_level_set_assembly_bc(self.h, _ptr(b), b.shape[0]))

    def assembly_state_size(self):
        n = ctypes.c_int64()
        self.ctx.check(self.ctx.lib.alfi_level_assembly_state_size(self.h, ctypes.byref(n)))
        return n.value

    def assemble_mult(self, nu, gamma, adv, state, x, y):
        """y = (nu K + gamma D + adv N(state)) x without boundary conditions, matrix-free (alfi_level_assemble_mult): the
        level's operator and patch factors stay as they are."""
        self.ctx.check(self.ctx.lib.alfi_level_assemble_mult(self.h, float(nu), float(gamma), float(adv),
                                                             state.ptr if state is not None else None, x.ptr, y.ptr))

    def set_supg(self, V, nq=None, cells=None):
        """Quadrature tables for the device-side SUPG terms (alfi_level_set_supg): the rule and tabulation of
        ``_hostlib.supg`` (degree 2k) and the cell sizes.  ``cells`` (partitioned level): the cells given to
        ``set_assembly``."""
        from . import _hostlib
        from .elements import simplex_quadrature
        el, d = V.element, V.dim
        deg = 3 if (el.bubble or el.degree == 3) else el.degree
        lam, wq = simplex_quadrature(d, nq or (deg + 1))
        phi, dphi = el.tabulate(lam)
        d2phi = el.tabulate_hessian(lam)
        h = _hostlib.cell_size(V.mesh)
        if cells is not None:
            h = np.asarray(h)[np.asarray(cells)]
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (wq, phi, dphi, d2phi, h)]
        self.ctx.check(self.ctx.lib.alfi_level_set_supg(self.h, len(wq), _ptr(arrs[0]), _ptr(arrs[1]), _ptr(arrs[2]),
                                                        _ptr(arrs[3]), _ptr(arrs[4])))

    def assemble_supg(self, nu, gamma, adv, state, weight, magic, apply_bc=True):
        """The refresh of a stabilised run in one pass (alfi_level_assemble_supg): A = nu K + gamma D + adv N(state) + the
        linearised SUPG term, then the boundary conditions."""
        self.ctx.check(self.ctx.lib.alfi_level_assemble_supg(self.h, float(nu), float(gamma), float(adv), state.ptr,
                                                             float(weight), float(magic), 1 if apply_bc else 0))

    def supg(self, nu, weight, magic, state, add_to_operator=True, F=None):
        """SUPG terms about ``state`` on the device (alfi_level_supg): the linearisation into the operator and / or the
        residual contribution added to the device vector ``F``."""
        self.ctx.check(self.ctx.lib.alfi_level_supg(self.h, float(nu), float(weight), float(magic), state.ptr,
                                                    1 if add_to_operator else 0, F.ptr if F is not None else None))

    def apply_bc(self):
        self.ctx.check(self.ctx.lib.alfi_level_apply_bc(self.h))

    def assemble(self, nu, gamma, adv, state=None, apply_bc=True):
        """A = nu K + gamma D + adv N(state) written into the level's operator on the device (alfi_level_assemble);
        ``state``: DeviceVec / RawVec with the level's nodal field.  The patches must be factored again afterwards."""
        self.ctx.check(self.ctx.lib.alfi_level_assemble(self.h, float(nu), float(gamma), float(adv),
                                                        state.ptr if state is not None else None, 1 if apply_bc else 0))

    def get_values(self):
        """The operator values in the host layout (nnzb, bs, bs)."""
        out = np.empty((self.nnzb, self.bs, self.bs))
        self.ctx.check(self.ctx.lib.alfi_level_get_values(self.h, _ptr(out)))
        return out

    def set_patches(self, patch_ptr, patch_dofs):
        pp = np.ascontiguousarray(patch_ptr, dtype=np.int64)
        pd = np.ascontiguousarray(patch_dofs, dtype=np.int32)
        self.ctx.check(self.ctx.lib.alfi_patches_set(self.h, len(pp) - 1, _ptr(pp), _ptr(pd)))

    def set_patch_groups(self, groups):
        """Condensed patch factors (alfi_patches_set_groups): one label per entry of patch_dofs, >= 0 = group of the patch the
        entry belongs to, -1 = skeleton; None = dense inverses."""
        g = None if groups is None else np.ascontiguousarray(groups, dtype=np.int32)
        self.ctx.check(self.ctx.lib.alfi_patches_set_groups(self.h, _ptr(g)))

    def factor_bytes(self):
        b = ctypes.c_int64()
        self.ctx.check(self.ctx.lib.alfi_patches_factor_bytes(self.h, ctypes.byref(b)))
        return b.value

    def set_partition_of_unity(self, on=True):
        self.ctx.check(self.ctx.lib.alfi_patches_set_partition_of_unity(self.h, 1 if on else 0))

    def set_multiplicative(self, iterset, symmetrise):
        """Multiplicative sweeps in the order ``iterset`` (None / empty = back to additive); returns the number of
        dependency wavefronts one sweep was scheduled into."""
        it = np.ascontiguousarray(iterset if iterset is not None else [], dtype=np.int64)
        self.ctx.check(self.ctx.lib.alfi_patches_set_multiplicative(self.h, len(it), _ptr(it), 1 if symmetrise else 0))
        nw = ctypes.c_int64()
        self.ctx.check(self.ctx.lib.alfi_patches_multiplicative_levels(self.h, ctypes.byref(nw)))
        return nw.value

    def factor(self):
        self.ctx.check(self.ctx.lib.alfi_patches_factor(self.h))

    def factor_with_fallback(self):
        """factor(); condensed factors that fail the residual probe are repaired in place (pivoted LU of their Schur
        complements, kernels_check.hip).  Only if they STILL fail beyond ALFI_PATCH_CHECK_FAIL -- or the repair does not
        apply (Schur complements above 4096 dofs) -- fall back to dense inverses for this level and factor again.  Returns
        True if it fell back."""
        try:
            self.factor()
            self._warn_if_flagged()
            return False
        except AlfiHipError as e:
            if "condensed" not in str(e):
                raise
        self.set_patch_groups(None)
        self.factor()
        self._warn_if_flagged()
        return True

    def _warn_if_flagged(self):
        """the setup only FAILS beyond 1000 x the probe tolerance; patches that stay above the tolerance after the
        pivoted repair (ill-conditioned, not mis-factored: LAPACK would return the same inverse) degrade the smoother
        silently unless somebody says so."""
        worst, flagged, repaired, after = self.patch_check()
        if flagged > repaired:
            import warnings
            warnings.warn("alfi_patches_factor: %d of %d flagged patch factors stay above the probe tolerance after the "
                          "pivoted repair (worst residual %.2e): the patch operators are ill-conditioned"
                          % (flagged - repaired, flagged, after))

    def patch_check(self):
        """(worst residual || A_p X_p e - e || of the fast inversion, patches flagged, patches repaired by the pivoted
        re-inversion, worst residual afterwards) of the last ``factor()``."""
        w, wa = ctypes.c_double(), ctypes.c_double()
        f, r = ctypes.c_int64(), ctypes.c_int64()
        self.ctx.check(self.ctx.lib.alfi_patches_check(self.h, ctypes.byref(w), ctypes.byref(f), ctypes.byref(r),
                                                       ctypes.byref(wa)))
        return w.value, f.value, r.value, wa.value

    def patch_stats(self):
        a, b, c = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        self.ctx.check(self.ctx.lib.alfi_patches_stats(self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    def patch_inverse(self, p, n):
        out = np.empty((n, n))
        self.ctx.check(self.ctx.lib.alfi_patch_get_inverse(self.h, int(p), _ptr(out)))
        return out

    def patch_apply(self, x, y):
        self.ctx.check(self.ctx.lib.alfi_patch_apply(self.h, x.ptr, y.ptr))

    def spmv(self, x, y):
        self.ctx.check(self.ctx.lib.alfi_spmv(self.h, x.ptr, y.ptr))

    def zero_bc(self, v):
        """v[Dirichlet dofs of the level] = 0 on the device (alfi_level_zero_bc)."""
        self.ctx.check(self.ctx.lib.alfi_level_zero_bc(self.h, v.ptr))

    def halo_forward(self, v):
        """Partitioned level: ghost slots of v <- their owners' values (no-op otherwise)."""
        self.ctx.check(self.ctx.lib.alfi_level_halo_forward(self.h, v.ptr))

    def halo_sum(self, v):
        """Partitioned level with a sum-exchange plan: every holder of a shared node gets the sum of all holders' values."""
        self.ctx.check(self.ctx.lib.alfi_level_halo_sum(self.h, v.ptr))

    def halo_reverse_add(self, v):
        """Partitioned level: ghost slots of v added onto their owners' entries (no-op otherwise)."""
        self.ctx.check(self.ctx.lib.alfi_level_halo_reverse_add(self.h, v.ptr))

    def residual(self, b, x, r):
        self.ctx.check(self.ctx.lib.alfi_residual(self.h, b.ptr, x.ptr, r.ptr))

    def smooth(self, k, b, x, nonzero_guess=True):
        self.ctx.check(self.ctx.lib.alfi_smooth_fgmres(self.h, int(k), b.ptr, x.ptr, 1 if nonzero_guess else 0))

    def set_coarse_inverse(self, inv):
        """inv: dense (n, n) numpy array (copied) or a device pointer (int) that stays owned by the caller."""
        if isinstance(inv, (int, np.integer)):
            self.ctx.check(self.ctx.lib.alfi_coarse_set_inverse(self.h, vp(int(inv)), 1))
        else:
            inv = np.ascontiguousarray(inv, dtype=np.float64)
            assert inv.shape == (self.n, self.n)
            self.ctx.check(self.ctx.lib.alfi_coarse_set_inverse(self.h, _ptr(inv), 0))

    def coarse_factor(self):
        """Dense inverse of this level's operator built by the library itself (blocked Gauss-Jordan on the FP64 matrix cores +
        residual probe, alfi_coarse_factor); returns the probe residual || A X e - e ||."""
        self.ctx.check(self.ctx.lib.alfi_coarse_factor(self.h))
        return self.coarse_residual()

    def coarse_factor_sparse(self, node_coords=None, leaf_nodes=0):
        """Sparse direct factorisation of this level's operator (multifrontal block L D U on a nested-dissection ordering,
        alfi_coarse_factor_sparse) for coarse grids whose dense inverse (8 n^2 bytes) is too large; ``node_coords``
        (nodes, dim) drives the geometric bisection (None: level sets of the graph).  Returns the probe residual."""
        if node_coords is None:
            self.ctx.check(self.ctx.lib.alfi_coarse_factor_sparse(self.h, None, 0, int(leaf_nodes)))
        else:
            xy = np.ascontiguousarray(node_coords, dtype=np.float64)
            assert xy.ndim == 2 and xy.shape[0] * self.bs == self.n, (xy.shape, self.n, self.bs)
            self.ctx.check(self.ctx.lib.alfi_coarse_factor_sparse(self.h, _ptr(xy), int(xy.shape[1]), int(leaf_nodes)))
        return self.coarse_residual()

    def coarse_factor_auto(self, node_coords=None, mode=None, geometric=False):
        """The coarse factorisation the front ends use: ``mode`` "dense", "sparse" or "auto" (sparse from
        ``coarse_sparse_min()`` dofs on).  The choice is remembered, so a later call without arguments (new operator values,
        every Newton step) repeats it.  The sparse path bisects by graph level sets; ``geometric``: by ``node_coords`` instead
        (measured on ldc3d coarse grids: level sets give 13-15 % less fill, hence the default)."""
        if mode is not None or node_coords is not None or not hasattr(self, "_coarse_choice"):
            self._coarse_choice = (mode or "auto", node_coords if geometric else None)
        mode, node_coords = self._coarse_choice
        if mode == "sparse" or (mode == "auto" and self.n >= coarse_sparse_min()):
            rc = self.coarse_factor_sparse(node_coords)
        else:
            rc = self.coarse_factor()
        res = self.coarse_residual()
        if res > 1e-5:         # (the setup fails only beyond ALFI_COARSE_CHECK_FAIL = 1e-2; rounds 1-2 failed at 1e-5)
            import warnings
            warnings.warn("coarse factorisation: probe residual || A X e - e || = %.2e" % res)
        return rc

    def coarse_factor_bytes(self):
        b = ctypes.c_int64()
        self.ctx.check(self.ctx.lib.alfi_coarse_factor_bytes(self.h, ctypes.byref(b)))
        return b.value

    def coarse_residual(self):
        """|| A X e - e ||_inf of the coarse inverse built by ``coarse_factor`` (-1 for a caller-supplied inverse)."""
        w = ctypes.c_double()
        self.ctx.check(self.ctx.lib.alfi_coarse_residual(self.h, ctypes.byref(w)))
        return w.value

    def coarse_solve(self, b, x):
        self.ctx.check(self.ctx.lib.alfi_coarse_solve(self.h, b.ptr, x.ptr))

    def close(self):
        if self.h:
            self.ctx.lib.alfi_level_destroy(self.h)
            self.h = None


class Transfer(object):
    def __init__(self, ctx, coarse, fine, T):
        """T: alfi_amd.problem.TransferData."""
        self.ctx, self.coarse, self.fine = ctx, coarse, fine
        keep = []
        P, PT = _bsr_struct(T.P, keep), _bsr_struct(T.PT, keep)
        PTp = _bsr_struct(T.PT_plain, keep) if T.PT_plain is not T.PT else None
        DI, DIT = _bsr_struct(T.D_I, keep), _bsr_struct(T.D_IT, keep)
        blk = np.ascontiguousarray(T.blk_dofs, dtype=np.int32)
        K = np.ascontiguousarray(T.K_II, dtype=np.float64)
        D = np.ascontiguousarray(T.D_II, dtype=np.float64)
        h = vp()
        ctx.check(ctx.lib.alfi_transfer_create(ctx.h, coarse.h, fine.h, ctypes.byref(P), ctypes.byref(PT),
                                               ctypes.byref(PTp) if PTp is not None else None, ctypes.byref(DI),
                                               ctypes.byref(DIT), blk.shape[0], blk.shape[1], _ptr(blk), _ptr(K),
                                               _ptr(D), ctypes.byref(h)))
        self.h = h
        inj = getattr(T, "inject_map", None)
        serial = not getattr(coarse, "n_own", None) and not getattr(fine, "n_own", None)
        if inj is not None and serial:
            inj = np.ascontiguousarray(inj, dtype=np.int32)
            ctx.check(ctx.lib.alfi_transfer_set_injection(h, _ptr(inj)))
        elif getattr(T, "inject_matrix", None) is not None and serial:
            # non-nested (barycentric) hierarchy: inject = point evaluation at the coarse nodes, a sparse matrix on the device
            import scipy.sparse as sp
            J = sp.csr_matrix(T.inject_matrix)
            J.sort_indices()
            rp = np.ascontiguousarray(J.indptr, dtype=np.int32)
            ci = np.ascontiguousarray(J.indices, dtype=np.int32)
            va = np.ascontiguousarray(J.data, dtype=np.float64)
            st = CsrHost(J.shape[0], J.shape[1], _ptr(rp), _ptr(ci), _ptr(va))
            ctx.check(ctx.lib.alfi_transfer_set_injection_matrix(h, ctypes.byref(st)))

    def inject(self, xf, xc):
        self.ctx.check(self.ctx.lib.alfi_inject(self.h, xf.ptr, xc.ptr))

    def update(self, nu, gamma):
        self.ctx.check(self.ctx.lib.alfi_transfer_update(self.h, float(nu), float(gamma)))

    def block_inverse(self, blk, m):
        out = np.empty((m, m))
        self.ctx.check(self.ctx.lib.alfi_transfer_get_block_inverse(self.h, int(blk), _ptr(out)))
        return out

    def prolong(self, xc, xf):
        self.ctx.check(self.ctx.lib.alfi_prolong(self.h, xc.ptr, xf.ptr))

    def restrict(self, rf, rc, robust=True):
        self.ctx.check(self.ctx.lib.alfi_restrict(self.h, rf.ptr, rc.ptr, 1 if robust else 0))

    def close(self):
        if self.h:
            self.ctx.lib.alfi_transfer_destroy(self.h)
            self.h = None


def coarse_inverse(A_bsr):
    """Dense inverse of the coarsest operator by host LAPACK -- an alternative to ``Level.coarse_factor`` for callers that
    want to supply their own (``Multigrid(..., coarse_inv=...)``); the default path does not use it."""
    A = A_bsr.to_scipy().toarray()
    return np.linalg.inv(A)


def condense_patches(L):
    """Whether the level's patch factors are stored condensed: the generator supplied group labels (macro-star patches of
    the Scott-Vogelius hierarchy, sv.macro_cell_groups) and ALFI_CONDENSE is not 0."""
    import os
    return getattr(L, "patch_groups", None) is not None and os.environ.get("ALFI_CONDENSE", "1") != "0"


def coarse_sparse_min():
    """Coarse grids from this many dofs on get the sparse factorisation by default (dense inverse: 8 n^2 bytes and one n x n
    GEMV per cycle; sparse: O(n^{4/3}) bytes and ~6 launches per tree height)."""
    import os
    return int(os.environ.get("ALFI_COARSE_SPARSE_MIN", 8192))


class Multigrid(object):
    """Device-resident PCMG (solver.py:359-379) built from alfi_amd.problem.build_hierarchy output."""

    def __init__(self, ctx, levels, transfers, k, robust_restriction=False, coarse_inv=None, verbose=False, coarse="auto"):
        """coarse: "dense" (explicit inverse, alfi_coarse_factor), "sparse" (multifrontal factors, alfi_coarse_factor_sparse)
        or "auto": sparse from COARSE_SPARSE_MIN dofs on (env ALFI_COARSE_SPARSE_MIN)."""
        import time
        t0 = time.time()
        dlevels = []
        for L in levels:
            dl = Level(ctx, L.A, L.bc_dofs)
            if L.level > 0:
                dl.set_patches(L.patch_ptr, L.patch_dofs)
                if condense_patches(L):
                    dl.set_patch_groups(L.patch_groups)
                dl.factor_with_fallback()
            elif coarse_inv is not None:
                dl.set_coarse_inverse(coarse_inv)          # an inverse supplied by the caller (numpy array / device pointer)
            else:                                          # the library's own factorisation, dense or multifrontal
                dl.coarse_factor_auto(getattr(getattr(L, "V", None), "node_coords", None), coarse)
            dlevels.append(dl)
        self._from_device_levels(ctx, dlevels, transfers, k, robust_restriction)
        if verbose:
            print("[alfi_amd] device hierarchy ready in %.1fs" % (time.time() - t0), flush=True)

    def _from_device_levels(self, ctx, dlevels, transfers, k, robust_restriction):
        """dlevels: hip.Level objects (level 0 with a coarse inverse, the others with factored patches)."""
        self.ctx = ctx
        self.levels, self.transfers = list(dlevels), []
        for i, T in enumerate(transfers):
            dt = Transfer(ctx, self.levels[i], self.levels[i + 1], T)
            dt.update(T.nu, T.gamma)
            self.transfers.append(dt)
        lv = (vp * len(self.levels))(*[l.h for l in self.levels])
        tr = (vp * max(1, len(self.transfers)))(*[t.h for t in self.transfers])
        h = vp()
        ctx.check(ctx.lib.alfi_mg_create(ctx.h, len(self.levels), lv, tr, int(k), 1 if robust_restriction else 0,
                                         ctypes.byref(h)))
        self.h = h
        self.k = k
        ctx.sync()

    def variant(self, robust_restriction):
        """A second cycle handle (alfi_mg) over the SAME device levels and transfers with the other restriction setting
        (`--restriction`, alfi/driver.py:41); ``close_handle()`` releases it without touching the shared levels."""
        other = Multigrid.__new__(Multigrid)
        other.ctx, other.levels, other.transfers, other.k = self.ctx, self.levels, self.transfers, self.k
        lv = (vp * len(self.levels))(*[l.h for l in self.levels])
        tr = (vp * max(1, len(self.transfers)))(*[t.h for t in self.transfers])
        h = vp()
        self.ctx.check(self.ctx.lib.alfi_mg_create(self.ctx.h, len(self.levels), lv, tr, int(self.k),
                                                   1 if robust_restriction else 0, ctypes.byref(h)))
        other.h = h
        return other

    def close_handle(self):
        if self.h:
            self.ctx.lib.alfi_mg_destroy(self.h)
            self.h = None

    def vcycle(self, b, x):
        self.ctx.check(self.ctx.lib.alfi_mg_vcycle(self.h, b.ptr, x.ptr))

    def fcycle(self, b, x):
        self.ctx.check(self.ctx.lib.alfi_mg_fcycle(self.h, b.ptr, x.ptr))

    def close(self):
        if self.h:
            self.ctx.lib.alfi_mg_destroy(self.h)
            self.h = None
        for t in self.transfers:
            t.close()
        for l in self.levels:
            l.close()


class RawVec(object):
    """A device buffer owned by someone else (a torch tensor, a slice of a DeviceVec) in the shape the wrappers expect
    (``.ptr``); with a ctx it can be filled and read like a DeviceVec."""

    def __init__(self, ptr, n, ctx=None, parent=None):
        self.ptr, self.n, self.ctx, self._parent = vp(int(ptr)), int(n), ctx, parent

    def set(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.shape == (self.n,)
        self.ctx.check(self.ctx.lib.alfi_memcpy_h2d(self.ctx.h, self.ptr, _ptr(a), self.n * 8))

    def get(self):
        out = np.empty(self.n, dtype=np.float64)
        self.ctx.check(self.ctx.lib.alfi_memcpy_d2h(self.ctx.h, _ptr(out), self.ptr, self.n * 8))
        return out


def view(v, off, n):
    """Entries [off, off + n) of a device vector as a vector of its own (no copy; keeps its parent alive)."""
    assert 0 <= off and off + n <= v.n
    return RawVec(v.ptr.value + 8 * int(off), n, ctx=getattr(v, "ctx", None), parent=v)


class Csr(object):
    """Scalar CSR matrix on the device (alfi_csr_*): the rank's rows of the discrete divergence and its transpose."""

    def __init__(self, ctx, M):
        import scipy.sparse as sp
        M = sp.csr_matrix(M)
        M.sort_indices()
        rp = np.ascontiguousarray(M.indptr, dtype=np.int32)
        ci = np.ascontiguousarray(M.indices, dtype=np.int32)
        va = np.ascontiguousarray(M.data, dtype=np.float64)
        st = CsrHost(M.shape[0], M.shape[1], _ptr(rp), _ptr(ci), _ptr(va))
        self.ctx, self.shape = ctx, M.shape
        h = vp()
        ctx.check(ctx.lib.alfi_csr_create(ctx.h, ctypes.byref(st), ctypes.byref(h)))
        self.h = h

    def mult(self, x, y, b=None, alpha=1.0, mode=0):
        """mode 0: y = M x;  1: y = b - alpha M x;  2: y += M x;  3: y = alpha M x"""
        self.ctx.check(self.ctx.lib.alfi_csr_mult(self.h, x.ptr, y.ptr, b.ptr if b is not None else None, float(alpha),
                                                  int(mode)))

    def close(self):
        if self.h:
            self.ctx.lib.alfi_csr_destroy(self.h)
            self.h = None


class Saddle(object):
    """Outer solve of one Newton step (alfi/solver.py:386-422): FGMRES around PCFIELDSPLIT-Schur-full with the device
    multigrid as fieldsplit_0 and DGMassInv (solver.py:15-38) as fieldsplit_1."""

    def __init__(self, mg, B, mass_diag, nu, gamma, remove_constant_nullspace=True, mass_inv=None, n_u=None):
        """mg: hip.Multigrid; B: scipy CSR (pressure dofs x velocity dofs); mass_diag: diagonal of the P0 mass matrix;
        mass_inv: scipy sparse inverse of a block-diagonal (discontinuous P_k) pressure mass matrix, replaces mass_diag.
        On a PARTITIONED finest level (alfi_amd.dist) B holds the rank's pressure rows over all local velocity dofs (owned
        + ghost), n_u = the owned velocity dofs (vectors are (owned velocity | owned pressure)) and the call is collective."""
        import scipy.sparse as sp
        self.mg, self.ctx = mg, mg.ctx
        B = sp.csr_matrix(B)
        B.sort_indices()
        BT = B.T.tocsr()
        BT.sort_indices()
        keep = []

        def st(M):
            rp = np.ascontiguousarray(M.indptr, dtype=np.int32)
            ci = np.ascontiguousarray(M.indices, dtype=np.int32)
            va = np.ascontiguousarray(M.data, dtype=np.float64)
            keep.extend([rp, ci, va])
            return CsrHost(M.shape[0], M.shape[1], _ptr(rp), _ptr(ci), _ptr(va))
        sB, sBT = st(B), st(BT)
        if mass_diag is None:
            mass_diag = np.ones(B.shape[0])
        md = np.ascontiguousarray(mass_diag, dtype=np.float64)
        h = vp()
        self.ctx.check(self.ctx.lib.alfi_saddle_create(mg.h, ctypes.byref(sB), ctypes.byref(sBT), _ptr(md), float(nu),
                                                       float(gamma), 1 if remove_constant_nullspace else 0,
                                                       ctypes.byref(h)))
        self.h = h
        self.n_u, self.n_p = (B.shape[1] if n_u is None else int(n_u)), B.shape[0]
        self.n = self.n_u + self.n_p
        if mass_inv is not None:
            Mi = sp.csr_matrix(mass_inv)
            Mi.sort_indices()
            sM = st(Mi)
            self.ctx.check(self.ctx.lib.alfi_saddle_set_mass_inverse(self.h, ctypes.byref(sM)))

    def update(self, nu, gamma):
        self.ctx.check(self.ctx.lib.alfi_saddle_update(self.h, float(nu), float(gamma)))

    def dot(self, x, y):
        """x . y of two (velocity | pressure) device vectors, summed over the ranks on partitioned levels (alfi_saddle_dot)."""
        out = ctypes.c_double()
        self.ctx.check(self.ctx.lib.alfi_saddle_dot(self.h, x.ptr, y.ptr, ctypes.byref(out)))
        return out.value

    def mult(self, x, y):
        self.ctx.check(self.ctx.lib.alfi_saddle_mult(self.h, x.ptr, y.ptr))

    def precond(self, x, y):
        self.ctx.check(self.ctx.lib.alfi_saddle_precond(self.h, x.ptr, y.ptr))

    def solve(self, b, x, rtol=1e-8, atol=1e-8, max_it=500, restart=30):
        """Returns (iterations, true residual norm)."""
        its, rn = ctypes.c_int(), ctypes.c_double()
        self.ctx.check(self.ctx.lib.alfi_saddle_solve(self.h, b.ptr, x.ptr, float(rtol), float(atol), int(max_it),
                                                      int(restart), ctypes.byref(its), ctypes.byref(rn)))
        return its.value, rn.value

    def close(self):
        if self.h:
            self.ctx.lib.alfi_saddle_destroy(self.h)
            self.h = None
