"""TEST INFRASTRUCTURE ONLY -- CPU (NumPy/SciPy) restatement of the reference's multigrid hot path.

PARITY UNPINNED: the reference (florianwechsung/alfi) ships no tests, golden vectors or logs, and its arithmetic
lives in un-vendored, un-pinned third-party code (PETSc PCPATCH / PCMG / KSPFGMRES / MatMult, Firedrake PatchPC and
prolong/restrict, PyOP2 par_loops) that is not installed here.  This oracle therefore follows
the reference's *call sites and data flow* (cited per function) plus the documented behaviour of those libraries; it
is pinned only by its own mathematical property tests (tests/test_host_generator.py, tests/test_graddiv.py), committed
fixtures produced by it (tests/golden/) and -- for the bubble transfer, the one place where the reference ships native
code -- by the reference's own C kernels compiled into oracle/_ref/ (oracle/build_ref.py, tests/test_ref_kernels.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The product path
(alfi_amd/) never does.
"""
import numpy as np
import scipy.linalg
import scipy.sparse as sp
import scipy.sparse.linalg as spla


# ---------------------------------------------------------------------------------------------------------------------
# S1/S2: vertex-star patches, literal restatement of alfi/relaxation.py:33-34, 110-160
# ---------------------------------------------------------------------------------------------------------------------
def star_patches_literal(V):
    """For every vertex: transitive support closure (vertex + incident edges, faces, cells; relaxation.py:33-34,
    158-160), then the dofs living on those points minus Dirichlet dofs (what PCPATCH keeps [3P]).  Pure-Python
    loops over the incidence relation -- small meshes only.  Returns list of (seed vertex, sorted dof array)."""
    m = V.mesh
    d = V.dim
    edges_of_v = [[] for _ in range(m.num_vertices)]
    for e, (a, b) in enumerate(m.edges):
        edges_of_v[a].append(e)
        edges_of_v[b].append(e)
    faces_of_v = [[] for _ in range(m.num_vertices)]
    if m.dim == 3:
        for f, vs in enumerate(m.faces):
            for a in vs:
                faces_of_v[a].append(f)
    out = []
    for v in range(m.num_vertices):
        nodes = [int(V.vertex_nodes[v])]
        if V.element.has_edge_nodes:
            nodes += [int(q) for e in edges_of_v[v] for q in np.atleast_1d(V.edge_nodes[e])]
        if V.element.has_face_nodes:
            nodes += [int(V.face_nodes[f]) for f in faces_of_v[v]]
        nodes = sorted(n for n in set(nodes) if not V.bc_node_mask[n])
        if nodes:
            out.append((v, np.array([n * d + c for n in nodes for c in range(d)], dtype=np.int32)))
    return out


# ---------------------------------------------------------------------------------------------------------------------
# operator: independent quadrature-based assembly of the velocity-block form (alfi/solver.py:565-568 linearised;
# alfi/transfer.py:319-324 for the symmetric transfer form)
# ---------------------------------------------------------------------------------------------------------------------
def assemble_form(V, nu=0.0, gamma=0.0, adv=0.0, wind=None, nq=6, gamma_full=0.0):
    """CSR of nu (2 sym grad u, grad v) + gamma (cell_avg div u, div v) + gamma_full (div u, div v)
    + adv ((w.grad)u + (u.grad)w, v), no BCs.  gamma: PkP0 forms (solver.py:565-568, transfer.py:319-332); gamma_full:
    Scott-Vogelius forms (solver.py:609-619, transfer.py:295-309)."""
    from alfi_amd.elements import simplex_quadrature
    m, d, el = V.mesh, V.dim, V.element
    lam, wq = simplex_quadrature(d, nq)
    phi, dphi = el.tabulate(lam)                                   # (q, a), (q, a, i)
    g, vol = m.cell_geometry()                                     # (c, i, x), (c,)
    nloc, nc = el.nloc, m.num_cells
    gradphi = np.einsum("qai,cix->cqax", dphi, g)                  # physical gradients (c, q, a, x)
    Ae = np.zeros((nc, nloc, d, nloc, d))
    if nu:
        G = np.einsum("q,cqax,cqbx->cab", wq, gradphi, gradphi)
        H = np.einsum("q,cqay,cqbx->caxby", wq, gradphi, gradphi)    # entry (a,x),(b,y) = int d_y phi_a d_x phi_b
        Ae += nu * vol[:, None, None, None, None] * H
        for x in range(d):
            Ae[:, :, x, :, x] += nu * vol[:, None, None] * G
    if gamma:
        bdiv = np.einsum("q,cqax->cax", wq, gradphi)                 # cell average of d_x phi_a
        Ae += gamma * vol[:, None, None, None, None] * np.einsum("cax,cby->caxby", bdiv, bdiv)
    if gamma_full:
        Ae += gamma_full * vol[:, None, None, None, None] * np.einsum("q,cqax,cqby->caxby", wq, gradphi, gradphi)
    if adv:
        wloc = wind[V.cell_nodes]                                    # (c, k, x)
        wq_ = np.einsum("qk,ckx->cqx", phi, wloc)                    # w at quadrature points
        gradw = np.einsum("cqkx,cky->cqyx", gradphi, wloc)           # gradw[c,q,y,x] = d_x w_y
        conv = np.einsum("cqx,cqbx->cqb", wq_, gradphi)              # (w . grad) phi_b
        N1 = np.einsum("q,cqb,qa->cab", wq, conv, phi)
        for x in range(d):
            Ae[:, :, x, :, x] += adv * vol[:, None, None] * N1
        N2 = np.einsum("q,qb,cqyx,qa->caybx", wq, phi, gradw, phi)   # (a,y),(b,x): phi_b d_x w_y phi_a
        Ae += adv * vol[:, None, None, None, None] * N2
    dofs = V.node_dofs(V.cell_nodes)                                 # (c, nloc*d)
    nd = nloc * d
    rows = np.repeat(dofs, nd, axis=1).ravel()
    cols = np.tile(dofs, (1, nd)).ravel()
    A = sp.csr_matrix((Ae.reshape(nc, -1).ravel(), (rows, cols)), shape=(V.num_dofs, V.num_dofs))
    A.sum_duplicates()
    return A


def supg_residual(V, U, nu, weight, magic, h, nq):
    """SUPG term of the momentum residual (alfi/stabilisation.py:57-60, 83-88; alfi/solver.py:204-223), PkP0 pairs (grad p = 0):
    F[a, i] = int weight beta (Lu)_i (u . grad phi_a),  Lu = -nu div(2 sym grad u) + (grad u) u,
    beta = (4 u.u / h^2 + magic (4 nu / h^2)^2)^(-1/2).  U: (num_nodes, dim); h: (ncell,) cell size.  Returns (num_dofs,)."""
    from alfi_amd.elements import simplex_quadrature
    m, d, el = V.mesh, V.dim, V.element
    lam, wq = simplex_quadrature(d, nq)
    phi, dphi = el.tabulate(lam)
    d2 = el.tabulate_hessian(lam)
    g, vol = m.cell_geometry()
    gphi = np.einsum("qai,cix->cqax", dphi, g)                         # grad phi_a
    hphi = np.einsum("qaik,cix,cky->cqaxy", d2, g, g)                  # Hessian of phi_a
    Uc = U[V.cell_nodes]                                               # (c, a, i)
    u = np.einsum("qa,cai->cqi", phi, Uc)
    Gu = np.einsum("cqax,cai->cqix", gphi, Uc)                         # d_x u_i
    lap = np.einsum("cqaxx,cai->cqi", hphi, Uc)
    gdiv = np.einsum("cqaij,caj->cqi", hphi, Uc)                       # d_i div u
    Lu = -nu * (lap + gdiv) + np.einsum("cqix,cqx->cqi", Gu, u)
    beta = (4.0 * np.einsum("cqi,cqi->cq", u, u) / h[:, None] ** 2 + magic * (4.0 * nu / h[:, None] ** 2) ** 2) ** -0.5
    s = np.einsum("cqx,cqax->cqa", u, gphi)
    Fe = np.einsum("q,c,cq,cqi,cqa->cai", wq, vol * weight, beta, Lu, s)
    F = np.zeros((V.num_nodes, d))
    np.add.at(F, V.cell_nodes, Fe)
    return F.ravel()


def apply_bcs_matrix(A, bc_dofs):
    """Rows and columns of Dirichlet dofs -> identity (firedrake.assemble(a, bcs=bcs) [3P])."""
    n = A.shape[0]
    keep = np.ones(n)
    keep[bc_dofs] = 0.0
    Dk = sp.diags(keep)
    return (Dk @ A @ Dk + sp.diags(1.0 - keep)).tocsr()


# ---------------------------------------------------------------------------------------------------------------------
# S5/S6: PatchPC / PCPATCH additive star smoother, dense explicit inverses (alfi/solver.py:318-328, 599-602)
# ---------------------------------------------------------------------------------------------------------------------
class PatchSmoother(object):
    def __init__(self, A, patch_ptr, patch_dofs, bc_dofs, local_type="additive", iterset=None, symmetrise=False,
                 partition_of_unity=False):
        """local_type / iterset / symmetrise: ``patch_pc_patch_local_type``, the iteration set returned by the patch
        constructor (relaxation.py:139-150) and ``patch_pc_patch_symmetrise_sweep`` (solver.py:322-324);
        partition_of_unity: ``patch_pc_patch_partition_of_unity`` (solver.py:321, always False in the reference): the
        additive sum of a dof is weighted by 1 / (number of patches holding it) [3P: PCPATCH dof_weights]."""
        self.pou = partition_of_unity
        self.A = sp.csr_matrix(A)
        self.patch_ptr, self.patch_dofs, self.bc_dofs = patch_ptr, patch_dofs, bc_dofs
        self.local_type, self.symmetrise = local_type, symmetrise
        self.iterset = np.arange(len(patch_ptr) - 1) if iterset is None else np.asarray(iterset)
        self.update()

    def update(self, A=None):
        """Patch operators are principal sub-blocks of the level operator inverted with
        LAPACK getrf/getri (patch_pc_patch_dense_inverse, solver.py:602)."""
        if A is not None:
            self.A = sp.csr_matrix(A)
        self.inv = []
        for p in range(len(self.patch_ptr) - 1):
            dofs = self.patch_dofs[self.patch_ptr[p]:self.patch_ptr[p + 1]]
            Ap = self.A[dofs][:, dofs].toarray()
            self.inv.append(np.linalg.inv(Ap))

    def apply(self, x):
        if self.local_type == "multiplicative":
            return self.apply_multiplicative(x)
        return self.apply_additive(x)

    def apply_multiplicative(self, x):
        """PCApply_PATCH with local_type multiplicative [3P]: patches are visited in iteration-set order (then, with
        symmetrise_sweep, once more in reverse order); each solve sees the residual left by all earlier ones --
        PCPATCH keeps a running local right-hand side and subtracts A[closure(p), p] dy_p after every solve
        (its 'matrix with artificial dofs'), which equals solving with r_p = (x - A y)_p since every row that couples
        to a star patch's dofs lies in the closure of the star.  y[bc] = x[bc] at the end."""
        y = np.zeros_like(x)
        r = x.copy()
        Acsc = self.A.tocsc()
        seq = list(self.iterset)
        if self.symmetrise:
            seq = seq + seq[::-1]
        for p in seq:
            dofs = self.patch_dofs[self.patch_ptr[p]:self.patch_ptr[p + 1]]
            dy = self.inv[p] @ r[dofs]
            y[dofs] += dy
            r -= Acsc[:, dofs] @ dy
        y[self.bc_dofs] = x[self.bc_dofs]
        return y

    def apply_additive(self, x):
        """y = sum_p R_p^T A_p^{-1} R_p x, no partition of unity (solver.py:321); y[bc] = x[bc]."""
        y = np.zeros_like(x)
        for p, Ainv in enumerate(self.inv):
            dofs = self.patch_dofs[self.patch_ptr[p]:self.patch_ptr[p + 1]]
            y[dofs] += Ainv @ x[dofs]
        if self.pou:
            y /= np.maximum(np.bincount(self.patch_dofs, minlength=len(x)), 1)
        y[self.bc_dofs] = x[self.bc_dofs]
        return y


# ---------------------------------------------------------------------------------------------------------------------
# S8: KSPFGMRES with classical Gram-Schmidt, exactly k iterations (alfi/solver.py:314-317; PETSc defaults App. C)
# ---------------------------------------------------------------------------------------------------------------------
def fgmres(A, M, b, x, k, nonzero_guess=True):
    """Right-preconditioned flexible GMRES(k), one cycle, ``ksp_convergence_test skip``.  A: matvec callable,
    M: preconditioner callable.  Returns the new iterate (x is not modified)."""
    r = b - A(x) if nonzero_guess else b.copy()
    beta = np.linalg.norm(r)
    if beta == 0.0:
        return x.copy()
    n = b.shape[0]
    Vb = np.zeros((k + 1, n))
    Z = np.zeros((k, n))
    H = np.zeros((k + 1, k))
    Vb[0] = r / beta
    cs, sn = np.zeros(k), np.zeros(k)
    grs = np.zeros(k + 1)
    grs[0] = beta
    its = 0
    for j in range(k):
        Z[j] = M(Vb[j])
        w = A(Z[j])
        h = Vb[:j + 1] @ w                       # classical Gram-Schmidt: all dots from the same w
        w = w - h @ Vb[:j + 1]
        H[:j + 1, j] = h
        tt = np.linalg.norm(w)
        H[j + 1, j] = tt
        its = j + 1
        # Givens update of the Hessenberg column (KSPFGMRESUpdateHessenberg)
        hcol = H[:j + 2, j].copy()
        for i in range(j):
            t = hcol[i]
            hcol[i] = cs[i] * t + sn[i] * hcol[i + 1]
            hcol[i + 1] = -sn[i] * t + cs[i] * hcol[i + 1]
        den = np.hypot(hcol[j], hcol[j + 1])
        if den == 0.0:
            break
        cs[j], sn[j] = hcol[j] / den, hcol[j + 1] / den
        grs[j + 1] = -sn[j] * grs[j]
        grs[j] = cs[j] * grs[j]
        hcol[j] = den
        hcol[j + 1] = 0.0
        H[:j + 2, j] = hcol
        if tt == 0.0:
            break
        Vb[j + 1] = w / tt
    # back substitution on the triangularised Hessenberg (KSPFGMRESBuildSoln)
    y = np.zeros(its)
    for i in range(its - 1, -1, -1):
        y[i] = (grs[i] - H[i, i + 1:its] @ y[i + 1:its]) / H[i, i]
    return x + y @ Z[:its]


# ---------------------------------------------------------------------------------------------------------------------
# T5/T6: Schoeberl prolongation / restriction, literal data flow of alfi/transfer.py:194-275
# ---------------------------------------------------------------------------------------------------------------------
class SchoeberlTransfer(object):
    def __init__(self, P, A_sym, gammaD, blk_dofs, skeleton_dofs, P_restrict=None):
        """P: standard transfer (fine x coarse, bubble-corrected where transfer.py:334-356 says so); A_sym: the
        symmetric form nu K + gamma D on the fine level (transfer.py:319-324); gammaD: gamma * D (bform,
        transfer.py:326-332); blk_dofs: (nblk, m) coarse-cell interior dofs (transfer.py:13-46); skeleton_dofs:
        Dirichlet set of fix_coarse_boundaries (transfer.py:121-158)."""
        self.P = sp.csr_matrix(P)
        self.gD = sp.csr_matrix(gammaD)
        self.blk_dofs, self.skel = blk_dofs, skeleton_dofs
        A = sp.csr_matrix(A_sym)
        # patch_sub_pc_type lu on each coarse-cell patch (transfer.py:100-113)
        self.lu = [scipy.linalg.lu_factor(A[d][:, d].toarray()) for d in blk_dofs]

    def _patch_apply(self, x):
        """PatchPC.apply with the coarse-cell patches: additive, disjoint; y[bc] = x[bc]."""
        y = np.zeros_like(x)
        for d, lu in zip(self.blk_dofs, self.lu):
            y[d] = scipy.linalg.lu_solve(lu, x[d])
        y[self.skel] = x[self.skel]
        return y

    def prolong(self, coarse):
        rhs = self.P @ coarse                                  # standard_transfer(coarse, rhs)   :247
        b = self.gD @ rhs                                      # assemble(bform, bcs=bcs)         :249
        b[self.skel] = 0.0
        tildeu = self._patch_apply(b)                          # ksp.pc.apply                     :254-257
        return rhs - tildeu                                    #                                  :259

    def restrict(self, fine):
        tildeu = fine.copy()                                   #                                  :265
        tildeu[self.skel] = 0.0                                # bcs.apply(tildeu)                :266
        rhs = self._patch_apply(tildeu)                        #                                  :267-270
        b = self.gD @ rhs                                      # assemble(bform) -- no bcs        :272
        rhs = fine - b                                         #                                  :274
        return self.P.T @ rhs                                  # standard_transfer(rhs, coarse)   :275


def bubble_prolong_literal(Vc, Vf, coarse):
    """alfi/bubble.py:233-265 executed step by step with the per-cell kernels (split :57-91, combine :125-147) and the
    divide-by-multiplicity steps (:243-244, :265), the facet rescaling (:29-39, 251-253) and nodal prolongation of the
    P1 and bubble parts (:256-257).  coarse: (num_dofs,) -> fine (num_dofs,)."""
    from alfi_amd.mesh import TET_FACES
    from alfi_amd.fespace import nodal_prolongation, _facet_normals
    from alfi_amd.elements import NodalElement
    mc, mf = Vc.mesh, Vf.mesh
    a = np.vstack([np.eye(4), np.zeros((4, 4))])                                         # bubble.py:64-71
    b = np.vstack([-(np.ones((4, 4)) - np.eye(4)) / 3.0, np.eye(4)])                     # bubble.py:73-80
    both = coarse.reshape(-1, 3)
    p1c = np.zeros((mc.num_vertices, 3))
    fbc = np.zeros((mc.num_faces, 3))
    cntp1, cntfb = np.zeros(mc.num_vertices), np.zeros(mc.num_faces)
    for c in range(mc.num_cells):
        loc = both[Vc.cell_nodes[c]]                      # (8, 3): 4 vertex nodes then 4 face nodes
        for kk in range(8):
            for i in range(4):
                p1c[mc.cells[c, i]] += a[kk, i] * loc[kk]
                fbc[mc.cell_faces[c, i]] += b[kk, i] * loc[kk]
        cntp1[mc.cells[c]] += 1
        cntfb[mc.cell_faces[c]] += 1
    p1c /= cntp1[:, None]
    fbc /= cntfb[:, None]
    nrm = _facet_normals(mc)
    fbc = fbc + (1.0 / 0.625 - 1.0) * (fbc * nrm).sum(axis=1)[:, None] * nrm
    p1 = NodalElement(3, 1, False)
    P1 = nodal_prolongation(Vc, Vf, p1, p1, mf.cells, mc.cells, mf.num_vertices, mc.num_vertices)
    p1f = P1 @ p1c

    class _FB(object):
        node_bary = Vc.element.node_bary[-4:]

        @staticmethod
        def tabulate(lam):
            return NodalElement._bubbles(np.atleast_2d(lam))
    PF = nodal_prolongation(Vc, Vf, _FB, _FB, mf.cell_faces, mc.cell_faces, mf.num_faces, mc.num_faces)
    fbf = PF @ fbc
    ac = np.hstack([np.eye(4), (np.ones((4, 4)) - np.eye(4)) / 3.0])                     # bubble.py:129-132
    bc_ = np.hstack([np.zeros((4, 4)), np.eye(4)])                                       # bubble.py:133-136
    fine = np.zeros((Vf.num_nodes, 3))
    cnt = np.zeros(Vf.num_nodes)
    for c in range(mf.num_cells):
        out = ac.T @ p1f[mf.cells[c]] + bc_.T @ fbf[mf.cell_faces[c]]
        fine[Vf.cell_nodes[c]] += out
        cnt[Vf.cell_nodes[c]] += 1
    fine /= cnt[:, None]
    return fine.ravel()


# ---------------------------------------------------------------------------------------------------------------------
# S9: PCMG V-cycle and full cycle (alfi/solver.py:359-379; PETSc PCMGMCycle_Private / PCMGFCycle_Private)
# ---------------------------------------------------------------------------------------------------------------------
class Multigrid(object):
    """levels[l]: dict(A=csr, smoother=PatchSmoother or None (l=0), bc=bc dofs); transfers[l-1]: object with
    prolong(coarse)->fine and restrict(fine)->coarse between level l-1 and l.  After every transfer the target's
    Dirichlet dofs are zeroed (firedrake.mg Interpolation.mult / multTranspose [3P])."""

    def __init__(self, levels, transfers, k):
        self.levels, self.transfers, self.k = levels, transfers, k
        self.coarse_lu = spla.splu(sp.csc_matrix(levels[0]["A"]))       # AssembledPC + LU, solver.py:369-378

    def smooth(self, l, b, x):
        L = self.levels[l]
        return fgmres(lambda v: L["A"] @ v, L["smoother"].apply, b, x, self.k, nonzero_guess=True)

    def prolong(self, l, xc):
        xf = self.transfers[l - 1].prolong(xc)
        xf[self.levels[l]["bc"]] = 0.0
        return xf

    def restrict(self, l, rf):
        rc = self.transfers[l - 1].restrict(rf)
        rc[self.levels[l - 1]["bc"]] = 0.0
        return rc

    def vcycle(self, l, b, x):
        if l == 0:
            return self.coarse_lu.solve(b)
        x = self.smooth(l, b, x)
        r = b - self.levels[l]["A"] @ x
        bc = self.restrict(l, r)
        xc = self.vcycle(l - 1, bc, np.zeros_like(bc))
        x = x + self.prolong(l, xc)
        return self.smooth(l, b, x)

    def fcycle(self, b):
        L = len(self.levels) - 1
        bs = [None] * (L + 1)
        bs[L] = b
        for l in range(L, 0, -1):
            bs[l - 1] = self.restrict(l, bs[l])
        x = np.zeros_like(bs[0])
        for l in range(L):
            x = self.vcycle(l, bs[l], x)
            x = self.prolong(l + 1, x)
        return self.vcycle(L, bs[L], x)


def build_oracle_mg(levels, transfers, k, schoeberl_restriction=False, local_type="additive", itersets=None,
                    symmetrise=False):
    """Oracle multigrid from the host generator's LevelData/TransferData (alfi_amd/problem.py).  itersets[l]: patch
    iteration order of level l for multiplicative sweeps (None = patch order)."""
    olev = []
    for L in levels:
        A = L.A.to_scipy().tocsr()
        its = None if itersets is None else itersets[L.level]
        sm = PatchSmoother(A, L.patch_ptr, L.patch_dofs, L.bc_dofs, local_type, its, symmetrise) \
            if L.level > 0 else None
        olev.append(dict(A=A, smoother=sm, bc=L.bc_dofs))
    otr = []
    for T, L in zip(transfers, levels[1:]):
        otr.append(oracle_transfer(T, L, schoeberl_restriction))
    return Multigrid(olev, otr, k)


class _TransferPair(object):
    def __init__(self, st, PT_plain, schoeberl_restriction):
        self.st, self.PT_plain, self.sr = st, PT_plain, schoeberl_restriction

    def prolong(self, xc):
        return self.st.prolong(xc)

    def restrict(self, rf):
        # alfi/solver.py:595: vtransfer.restrict if self.restriction else firedrake's plain restrict
        return self.st.restrict(rf) if self.sr else self.PT_plain @ rf


def oracle_transfer(T, L, schoeberl_restriction=False):
    from alfi_amd.fespace import skeleton_node_mask
    V = L.V
    K = assemble_form(V, nu=1.0)
    if getattr(T, "skeleton_dofs", None) is not None:
        # Scott-Vogelius transfer on a bary hierarchy: full grad-div form, Dirichlet set = coarse MACRO facets
        D = assemble_form(V, gamma_full=1.0)
        skel = T.skeleton_dofs
    else:
        D = assemble_form(V, gamma=1.0)
        skel = np.flatnonzero(np.repeat(skeleton_node_mask(V), V.dim))
    st = SchoeberlTransfer(T.P.to_scipy(), T.nu * K + T.gamma * D, T.gamma * D, T.blk_dofs, skel)
    return _TransferPair(st, T.PT_plain.to_scipy().tocsr(), schoeberl_restriction)


# ---------------------------------------------------------------------------------------------------------------------
# outer solve of one Newton step: KSPFGMRES + PCFIELDSPLIT Schur (full factorisation) with fieldsplit_0 = PCMG full cycle
# and fieldsplit_1 = DGMassInv (alfi/solver.py:15-38, 386-422, 463-499)
# ---------------------------------------------------------------------------------------------------------------------
def saddle_solve(mg, A, B, mass_diag, nu, gamma, b, rtol=1e-8, atol=1e-8, max_it=500, restart=30,
                 remove_constant_nullspace=True, mass_inv=None):
    """[A B^T; B 0] x = b by right-preconditioned FGMRES(restart), classical Gram-Schmidt, zero initial guess, KSP's
    default convergence test on the recurrence residual.  Preconditioner (PCApply_FieldSplit_Schur, FULL [3P]):
    y_u = MG(b_u); y_p = -(nu + gamma) M_p^-1 (b_p - B y_u)  (DGMassInv.apply, solver.py:26-35); y_u = MG(b_u - B^T y_p);
    constants removed from y_p (the nullspace attached for enclosed flows).  Returns (x, iterations, residual history)."""
    nu_ = A.shape[0]
    n = nu_ + B.shape[0]
    # mass_inv: sparse inverse of a block-diagonal (discontinuous P_k) pressure mass matrix, replaces mass_diag
    minv = sp.diags(1.0 / np.asarray(mass_diag)) if mass_inv is None else sp.csr_matrix(mass_inv)

    def K(x):
        return np.concatenate([A @ x[:nu_] + B.T @ x[nu_:], B @ x[:nu_]])

    def P(v):
        yu = mg.fcycle(v[:nu_])
        yp = -(nu + gamma) * (minv @ (v[nu_:] - B @ yu))
        yu = mg.fcycle(v[:nu_] - B.T @ yp)
        if remove_constant_nullspace:
            yp = yp - yp.mean()
        return np.concatenate([yu, yp])
    x = np.zeros(n)
    r = b.copy()
    bnorm = np.linalg.norm(b)
    tol = max(rtol * bnorm, atol)
    hist = [bnorm]
    its = 0
    converged = bnorm <= tol
    while not converged and its < max_it:
        beta = np.linalg.norm(r)
        V = np.zeros((restart + 1, n))
        Z = np.zeros((restart, n))
        H = np.zeros((restart + 1, restart))
        cs, sn, grs = np.zeros(restart), np.zeros(restart), np.zeros(restart + 1)
        V[0], grs[0] = r / beta, beta
        j = 0
        while j < restart and its < max_it:
            Z[j] = P(V[j])
            w = K(Z[j])
            h = V[:j + 1] @ w
            w = w - h @ V[:j + 1]
            tt = np.linalg.norm(w)
            hcol = np.concatenate([h, [tt]])
            for i in range(j):
                t = hcol[i]
                hcol[i] = cs[i] * t + sn[i] * hcol[i + 1]
                hcol[i + 1] = -sn[i] * t + cs[i] * hcol[i + 1]
            den = np.hypot(hcol[j], hcol[j + 1])
            cs[j], sn[j] = hcol[j] / den, hcol[j + 1] / den
            grs[j + 1] = -sn[j] * grs[j]
            grs[j] = cs[j] * grs[j]
            hcol[j], hcol[j + 1] = den, 0.0
            H[:j + 2, j] = hcol
            its += 1
            hist.append(abs(grs[j + 1]))
            j += 1
            if abs(grs[j]) <= tol:
                converged = True
                break
            if j < restart:
                V[j] = w / tt
        y = np.zeros(j)
        for i in range(j - 1, -1, -1):
            y[i] = (grs[i] - H[i, i + 1:j] @ y[i + 1:j]) / H[i, i]
        x = x + y @ Z[:j]
        if converged:
            break
        r = b - K(x)
        converged = np.linalg.norm(r) <= tol
    return x, its, hist
