"""TEST INFRASTRUCTURE ONLY -- SPMD (one call per rank) NumPy restatement of the mesh-partitioned cycle.

PARITY UNPINNED (see oracle/alfi_oracle.py).  Restates what the reference gets from PETSc when every level's mesh is
distributed with a vertex-star overlap (alfi/solver.py:604-605), patches exist for owned vertices only
(alfi/relaxation.py:120-121) and PCPATCH / MatMult / VecMDot exchange ghosts and reduce over MPI [3P]: forward halo
before every operator or patch apply, reverse-add halo after the patch apply, all-reduced dot products.  It runs on the
rank-local data produced by alfi_amd.dist.localize (the product's own partitioner) and exchanges through
alfi_amd.dist.Comm on CPU tensors (gloo), so a world-size-2 CPU run checks partition + halo plans + exchange order
against the serial oracle.  The HIP library performs the same sequence of exchanges (csrc/api_level.hip, api_smoother.hip, api_cycles.hip).

Only tests/ import this module.
"""
import numpy as np
import scipy.sparse.linalg as spla
import scipy.sparse as sp
import torch


class DistOracle(object):
    def __init__(self, llev, ltr, lmin, k, comm, robust=False, mult_orders=None, symmetrise=False):
        """mult_orders[i]: iteration set (local patch indices) of local level i for multiplicative sweeps, or None for the
        additive smoother.  Under MPI PCPATCH sweeps over the rank's own patches with its local vectors and combines
        ranks additively [3P]."""
        self.lev, self.tr, self.lmin, self.k, self.comm, self.robust = llev, ltr, lmin, k, comm, robust
        self.mult_orders, self.symmetrise = mult_orders, symmetrise
        self.A = [L.A.to_scipy().tocsr() for L in llev]
        self.inv = []
        for L, A in zip(llev, self.A):
            invs = []
            for p in range(len(L.patch_ptr) - 1):
                d = L.patch_dofs[L.patch_ptr[p]:L.patch_ptr[p + 1]]
                invs.append(np.linalg.inv(A[d][:, d].toarray()))
            self.inv.append(invs)
        self.coarse_lu = None
        if lmin == 0 and llev[0].part.nb_own > 0:
            self.coarse_lu = spla.splu(sp.csc_matrix(self.A[0]))
        self.binv = []
        for T in ltr:
            self.binv.append([np.linalg.inv(T.nu * K + T.gamma * D) for K, D in zip(T.K_II, T.D_II)])
        self.mats = [dict((name, getattr(T, name).to_scipy().tocsr()) for name in ("P", "PT", "PT_plain", "D_I", "D_IT"))
                     for T in ltr]

    # -- exchanges: collective over all ranks (all_to_all), so the CALLER decides from rank-independent facts whether one
    #    happens: inside the smoother iff the level is distributed, inside a transfer iff its fine level is distributed --
    def halo_fwd(self, i, v):
        p, bs = self.lev[i].part, self.lev[i].bs
        sn = np.concatenate(p.send_nodes) if p.send_counts.sum() else np.zeros(0, dtype=np.int64)
        send = torch.from_numpy(np.ascontiguousarray(v.reshape(-1, bs)[sn].ravel()))
        recv = torch.zeros(p.nb_ghost * bs, dtype=torch.float64)
        self.comm.exchange(send, recv, p.send_counts * bs, p.recv_counts * bs)
        v[p.nb_own * bs:] = recv.numpy()

    def halo_rev(self, i, v):
        p, bs = self.lev[i].part, self.lev[i].bs
        sn = np.concatenate(p.send_nodes) if p.send_counts.sum() else np.zeros(0, dtype=np.int64)
        send = torch.from_numpy(np.ascontiguousarray(v[p.nb_own * bs:]))
        recv = torch.zeros(sn.shape[0] * bs, dtype=torch.float64)
        self.comm.exchange(send, recv, p.recv_counts * bs, p.send_counts * bs)
        np.add.at(v.reshape(-1, bs), sn, recv.numpy().reshape(-1, bs))

    def allsum(self, i, a):
        a = np.atleast_1d(np.asarray(a, dtype=np.float64)).copy()
        if self.lev[i].part.distributed:
            t = torch.from_numpy(a)
            self.comm.allreduce(t)
        return a

    # -- level operations (all take / return local-length vectors; only the owned prefix of results is meaningful) ------
    def spmv(self, i, x):
        L = self.lev[i]
        if L.part.distributed:
            self.halo_fwd(i, x)
        y = np.zeros(L.n)
        y[:L.n_own] = (self.A[i] @ x)[:L.n_own]
        return y

    def patch_apply(self, i, x):
        L = self.lev[i]
        if L.part.distributed:
            self.halo_fwd(i, x)
        y = np.zeros(L.n)
        order = None if self.mult_orders is None else self.mult_orders[i]
        if order is None:
            for p, Ainv in enumerate(self.inv[i]):
                d = L.patch_dofs[L.patch_ptr[p]:L.patch_ptr[p + 1]]
                y[d] += Ainv @ x[d]
        else:
            seq = list(order) + (list(order[::-1]) if self.symmetrise else [])
            for p in seq:                                     # local Gauss-Seidel: r_p = (x - A_loc y)_p
                d = L.patch_dofs[L.patch_ptr[p]:L.patch_ptr[p + 1]]
                y[d] += self.inv[i][p] @ (x[d] - (self.A[i][d] @ y))
        if L.part.distributed:
            self.halo_rev(i, y)
        y[L.bc_dofs] = x[L.bc_dofs]
        return y

    def smooth(self, i, b, x):
        """FGMRES(k), classical Gram-Schmidt, no convergence test (oracle.alfi_oracle.fgmres with global reductions)."""
        L, k = self.lev[i], self.k
        no = L.n_own
        r = np.zeros(L.n)
        r[:no] = b[:no] - self.spmv(i, x)[:no]
        beta = np.sqrt(self.allsum(i, r[:no] @ r[:no])[0])
        if beta == 0.0:
            return x
        V, Z = np.zeros((k + 1, L.n)), np.zeros((k, L.n))
        H = np.zeros((k + 1, k))
        V[0, :no] = r[:no] / beta
        cs, sn, grs = np.zeros(k), np.zeros(k), np.zeros(k + 1)
        grs[0] = beta
        for j in range(k):
            Z[j] = self.patch_apply(i, V[j])
            w = self.spmv(i, Z[j])
            h = self.allsum(i, V[:j + 1, :no] @ w[:no])
            w[:no] -= h @ V[:j + 1, :no]
            tt = np.sqrt(self.allsum(i, w[:no] @ w[:no])[0])
            hcol = np.concatenate([h, [tt]])
            for q in range(j):
                t = hcol[q]
                hcol[q] = cs[q] * t + sn[q] * hcol[q + 1]
                hcol[q + 1] = -sn[q] * t + cs[q] * hcol[q + 1]
            den = np.hypot(hcol[j], hcol[j + 1])
            cs[j], sn[j] = hcol[j] / den, hcol[j + 1] / den
            grs[j + 1] = -sn[j] * grs[j]
            grs[j] = cs[j] * grs[j]
            hcol[j], hcol[j + 1] = den, 0.0
            H[:j + 2, j] = hcol
            if j + 1 < k:
                V[j + 1, :no] = w[:no] / tt
        y = np.zeros(k)
        for q in range(k - 1, -1, -1):
            y[q] = (grs[q] - H[q, q + 1:k] @ y[q + 1:k]) / H[q, q]
        out = x.copy()
        out[:no] += y @ Z[:, :no]
        return out

    # -- transfers (index t links local levels t and t+1) ---------------------------------------------------------------
    def prolong(self, t, xc):
        T, M = self.tr[t], self.mats[t]
        F = self.lev[t + 1]
        if F.part.distributed:
            self.halo_fwd(t, xc)
        xf = np.zeros(F.n)
        xf[:F.n_own] = M["P"] @ xc
        if F.part.distributed:
            self.halo_fwd(t + 1, xf)
        bI = M["D_I"] @ xf
        m = T.blk_dofs.shape[1]
        for b, Binv in enumerate(self.binv[t]):
            xf[T.blk_dofs[b]] -= T.gamma * (Binv @ bI[b * m:(b + 1) * m])
        xf[F.bc_dofs] = 0.0
        return xf

    def restrict(self, t, rf):
        T, M = self.tr[t], self.mats[t]
        F, C = self.lev[t + 1], self.lev[t]
        if self.robust:
            if F.part.distributed:
                self.halo_fwd(t + 1, rf)
            m = T.blk_dofs.shape[1]
            tI = np.concatenate([Binv @ rf[T.blk_dofs[b]] for b, Binv in enumerate(self.binv[t])]) \
                if len(self.binv[t]) else np.zeros(0)
            s = rf[:F.n_own] - T.gamma * (M["D_IT"] @ tI)
            rc = M["PT"] @ s
        else:
            rc = M["PT_plain"] @ rf[:F.n_own]
        rc = np.asarray(rc).copy()
        if F.part.distributed:
            self.halo_rev(t, rc)
        rc[C.bc_dofs] = 0.0
        return rc

    # -- cycles ---------------------------------------------------------------------------------------------------------------
    def vcycle(self, i, b, x):
        L = self.lev[i]
        active = L.part.nb_own > 0
        if i == 0:
            if active and self.coarse_lu is not None:
                return self.coarse_lu.solve(b)
            return x
        if active:
            x = self.smooth(i, b, x)
            r = np.zeros(L.n)
            r[:L.n_own] = b[:L.n_own] - self.spmv(i, x)[:L.n_own]
        else:
            r = np.zeros(L.n)
        bc = self.restrict(i - 1, r)
        xc = self.vcycle(i - 1, bc, np.zeros(self.lev[i - 1].n))
        corr = self.prolong(i - 1, xc)
        if active:
            x = x.copy()
            x[:L.n_own] += corr[:L.n_own]
            x = self.smooth(i, b, x)
        return x

    def fcycle(self, b):
        top = len(self.lev) - 1
        bs = [None] * (top + 1)
        bs[top] = b
        for i in range(top, 0, -1):
            bs[i - 1] = self.restrict(i - 1, bs[i].copy())
        x = np.zeros(self.lev[0].n)
        for i in range(top):
            x = self.vcycle(i, bs[i], x)
            x = self.prolong(i, x)
        return self.vcycle(top, bs[top], x)


def dist_saddle_solve(mg, Bloc, mass_diag_loc, np_global, nu, gamma, b, rtol=1e-8, atol=1e-8, max_it=500, restart=30,
                      remove_constant_nullspace=True):
    """SPMD restatement of oracle.alfi_oracle.saddle_solve on the rank-local data of alfi_amd.dist.localize_pressure: owned
    velocity dofs | owned pressure dofs, every dot product all-reduced.  mg: DistOracle.  b: (n_own + np_own,)."""
    i = len(mg.lev) - 1
    L = mg.lev[i]
    n_own, n_loc = L.n_own, L.n
    np_own = Bloc.shape[0]
    n = n_own + np_own
    minv = 1.0 / np.asarray(mass_diag_loc)

    def allsum(a):
        return mg.allsum(i, np.atleast_1d(np.asarray(a, dtype=np.float64)))

    def K(x):
        u = np.zeros(n_loc)
        u[:n_own] = x[:n_own]
        Au = mg.spmv(i, u)                                  # fills the ghosts of u
        t = Bloc.T @ x[n_own:]
        mg.halo_rev(i, t)
        return np.concatenate([Au[:n_own] + t[:n_own], Bloc @ u])

    def P(v):
        bu = np.zeros(n_loc)
        bu[:n_own] = v[:n_own]
        yu = mg.fcycle(bu)
        mg.halo_fwd(i, yu)
        yp = -(nu + gamma) * minv * (v[n_own:] - Bloc @ yu)
        t = Bloc.T @ yp
        mg.halo_rev(i, t)
        bu2 = np.zeros(n_loc)
        bu2[:n_own] = v[:n_own] - t[:n_own]
        yu = mg.fcycle(bu2)
        if remove_constant_nullspace:
            yp = yp - allsum(yp.sum())[0] / np_global
        return np.concatenate([yu[:n_own], yp])

    x = np.zeros(n)
    r = b.copy()
    bnorm = float(np.sqrt(allsum(r @ r)[0]))
    tol = max(rtol * bnorm, atol)
    its, rnorm = 0, bnorm
    while rnorm > tol and its < max_it:
        V = np.zeros((restart + 1, n))
        Z = np.zeros((restart, n))
        H = np.zeros((restart + 1, restart))
        cs, sn, grs = np.zeros(restart), np.zeros(restart), np.zeros(restart + 1)
        V[0], grs[0] = r / rnorm, rnorm
        j = 0
        while j < restart and its < max_it:
            Z[j] = P(V[j])
            w = K(Z[j])
            h = allsum(V[:j + 1] @ w)
            w = w - h @ V[:j + 1]
            tt = float(np.sqrt(allsum(w @ w)[0]))
            hcol = np.concatenate([h, [tt]])
            for q in range(j):
                t = hcol[q]
                hcol[q] = cs[q] * t + sn[q] * hcol[q + 1]
                hcol[q + 1] = -sn[q] * t + cs[q] * hcol[q + 1]
            den = np.hypot(hcol[j], hcol[j + 1])
            cs[j], sn[j] = hcol[j] / den, hcol[j + 1] / den
            grs[j + 1] = -sn[j] * grs[j]
            grs[j] = cs[j] * grs[j]
            hcol[j], hcol[j + 1] = den, 0.0
            H[:j + 2, j] = hcol[:j + 2]
            its += 1
            rnorm = abs(grs[j + 1])
            j += 1
            if rnorm <= tol or tt == 0.0:
                break
            V[j] = w / tt
        y = np.linalg.solve(np.triu(H[:j, :j]), grs[:j])
        x = x + y @ Z[:j]
        r = b - K(x)
        rnorm = float(np.sqrt(allsum(r @ r)[0]))
    return x, its, rnorm
