"""The solve loop around the hot path: Newton on the stationary Navier-Stokes equations with Reynolds continuation, as
``alfi.solver.NavierStokesSolver.solve`` / ``alfi.driver.run_solver`` organise it (alfi/solver.py:257-300,
alfi/driver.py:95-128), every linear solve on the GPU.

Per Newton step (``snes_type newtonls``, basic line search, solver.py:463-472):

1. host: rediscretise the velocity block about the current velocity on every level -- the state reaches the coarse
   levels by ``inject`` (solver.py:595; here the index map of ``fespace.injection_map``) -- and hand the new values to
   the device hierarchy (``HipPatchPC.update`` -> re-gather and re-invert every patch, new coarse inverse);
2. host: nonlinear residual F(u, p) of solver.py:565-568 with the [P_k(+FB)]^d - P0 pair,
   F_u = A0 u + 1/2 N(u) u + B^T p,  F_p = B u   (N(u) v = (u.grad) v + (v.grad) u, so N(u) u = 2 (u.grad) u);
3. device: J d = -F by ``alfi_saddle_solve`` (FGMRES + fieldsplit Schur full, PCMG full cycles, DGMassInv).

The generator plays Firedrake's role (assembly, setup time); the arithmetic of the solves is libalfi_hip.so's.
"""
import time

import numpy as np

from . import _hostlib, hip
from .problem import BSR, build_hierarchy, build_pressure_coupling
from .solver import HipMG, mg_levels_solver, fieldsplit_0_mg, outer_solver


def _assemble(L, nu, gamma, adv, wind, with_bc, full_div=False):
    """full_div: the Scott-Vogelius grad-div term gamma (div u, div v) (solver.py:616) instead of the cell-averaged one."""
    V = L.V
    g, vol = V.mesh.cell_geometry()
    tens = V.element.reference_tensors()
    A = _hostlib.assemble_bsr(V.cell_nodes, g, vol, tens, V.dim, L.A.rowptr, L.A.colidx, nu=nu,
                              gamma=0.0 if full_div else gamma, gamma_full=gamma if full_div else 0.0, adv=adv,
                              wind=wind if adv else None)
    if with_bc:
        _hostlib.apply_bc_bsr(V.num_nodes, V.dim, L.A.rowptr, L.A.colidx, A, np.repeat(V.bc_node_mask, V.dim))
    return A


class HipNavierStokesSolver(object):
    """``solve(re)`` mirrors NavierStokesSolver.solve (solver.py:257-300): returns (z, info_dict) with the reference's
    keys Re, nu, linear_iter, nonlinear_iter, time."""

    def __init__(self, problem, nref, k, gamma=1e4, smoothing=None, restriction=False, ctx=None, verbose=False,
                 snes_rtol=None, snes_atol=None, snes_stol=1e-6, snes_max_it=20, discretisation="pkp0", stabilisation_type=None,
                 stabilisation_weight=None, supg_magic=9.0, device_assembly=None):
        """discretisation: "pkp0" ([P_k(+FB)]^d - P0 on the uniform hierarchy, ConstantPressureSolver solver.py:561-602) or
        "sv" ([P_k]^d - P_{k-1}^dg on the barycentric hierarchy with macro-star patches, ScottVogeliusSolver :604-662).
        device_assembly: refresh the level operators of every Newton step ON THE DEVICE (alfi_level_assemble: what
        PatchPC.update does inside PCPATCH, solver.py:320, 325) instead of rediscretising on the host and re-uploading;
        default: on (viscous, grad-div, advection and SUPG terms), unless ALFI_DEVICE_ASSEMBLY=0."""
        import os
        self.problem, self.gamma, self.verbose = problem, float(gamma), verbose
        if device_assembly is None:
            device_assembly = os.environ.get("ALFI_DEVICE_ASSEMBLY", "1") != "0"
        self.device_assembly = bool(device_assembly)
        self.timings = {"assemble_s": 0.0, "factor_s": 0.0, "residual_s": 0.0, "solve_s": 0.0, "newton_steps": 0}
        self._ctx_arg = ctx
        dim = problem.dim
        self.sv = discretisation == "sv"
        # stabilisation (solver.py:56-58, 66-68, 204-234): SUPG with the Shakib-Hughes-Johan coefficient (supg_method
        # "shakib", the default), default weight 0.1 in 3-D and 1 in 2-D (stabilisation.py:52-54), supg_magic 9
        if stabilisation_type in ("none", None):
            stabilisation_type = None
        if stabilisation_type not in (None, "supg"):
            raise NotImplementedError("stabilisation type %r (built: supg for the P0-pressure pairs)" % stabilisation_type)
        if stabilisation_type == "supg" and self.sv:
            raise NotImplementedError("supg with a discontinuous P_k pressure couples grad p into the momentum block")
        self.supg = stabilisation_type == "supg"
        self.supg_weight = float(stabilisation_weight) if stabilisation_weight is not None else (0.1 if dim == 3 else 1.0)
        self.supg_magic = float(supg_magic)
        self.char_L, self.char_U = problem.char_length(), problem.char_velocity()
        self.nullspace = bool(problem.has_nullspace())
        # hierarchy and device objects are created once (Stokes operator); values are replaced per Newton step
        self._values_on_device = False
        if self.sv:
            from .sv import build_sv_hierarchy, build_sv_pressure_coupling
            self.levels, self.transfers = build_sv_hierarchy(problem, nref, k, Re=0.0, gamma=gamma)
        else:
            # with the device-side refresh nobody reads a host copy of the operators: the generator then delivers the sparsity
            # only and the first (Stokes) operator is formed on the device as well (no host assembly, no 8 GB upload at config 4)
            self._values_on_device = self.device_assembly and self._device_assembly_possible()
            self.levels, self.transfers = build_hierarchy(problem, nref, k, Re=0.0, gamma=gamma, lazy=self._lazy_generation(),
                                                          operator_values=not self._values_on_device)
        if self.sv:      # patch = macro with the problem's relaxation direction (solver.py:339-342), sparse-LU patch options
            from .solver import configure_patch_solver_sv
            mgl = configure_patch_solver_sv(mg_levels_solver(dim, patch="macro", smoothing=smoothing,
                                                             relaxation_direction=problem.relaxation_direction()), dim)
        else:
            mgl = mg_levels_solver(dim, smoothing=smoothing)
        self.params = outer_solver(dim, fieldsplit_0_mg(mgl))
        L = self.levels[-1]
        self.nu = self.char_L * self.char_U
        self.Minv = None
        if self.sv:
            # B with the Dirichlet columns zeroed (the Jacobian's) and with all columns (the residual's), one pass
            self.B, self.B_raw, M, self.Minv = build_sv_pressure_coupling(L, both=True)
            self.vol = np.asarray(M.sum(axis=1)).ravel()                    # int psi_j: weights of the pressure integral
        else:
            # Dirichlet columns zeroed: the Jacobian's B; all columns: the residual's B
            self.B, self.B_raw, self.vol = build_pressure_coupling(L, both=True)
        self._create_device(restriction)
        self._asm_ready = False
        if self.device_assembly and not self._device_assembly_possible():
            self.device_assembly = False
        if self.device_assembly:
            failure = None
            try:
                self._setup_device_assembly()
            except hip.AlfiHipError as e:
                failure = e
            # (partitioned: the outcome is agreed over the ranks -- one rank on the host path and the others on the device path
            # would call different collectives)
            if self._any_rank(failure is not None):
                # (e.g. ALFI_SPMV=legacy: the refresh writes the lane-major operator layout) -- ADVICE r3: fall back, say so
                import warnings
                warnings.warn("device-side operator refresh not available (%s): the operators of every Newton step are "
                              "assembled on the host" % (failure if failure is not None else "another rank failed to set it up",))
                self.device_assembly = False
        if self._values_on_device:
            if self.device_assembly:
                self._first_operators_on_device()
            else:                           # the refresh could not be set up after all: the host assembler's Stokes operators
                for Lv in self.levels:
                    Lv.A = BSR(Lv.A.nbrows, Lv.A.nbcols, Lv.bs, Lv.A.rowptr, Lv.A.colidx,
                               _assemble(Lv, self.nu, self.gamma, 0.0, None, True, full_div=self.sv))
                self._values_on_device = False
                self._push_operators()
        self.rtol, self.atol = self.params["ksp_rtol"], self.params["ksp_atol"]
        tol2, tol3 = (1e-9, 1e-8), (1e-8, 1e-8)                            # snes_rtol / snes_atol, solver.py:484-499
        self.snes_rtol = snes_rtol if snes_rtol is not None else (tol2 if dim == 2 else tol3)[0]
        self.snes_atol = snes_atol if snes_atol is not None else (tol2 if dim == 2 else tol3)[1]
        self.snes_stol = snes_stol                                          # solver.py:490, 498
        self.snes_max_it = snes_max_it
        self.n_u, self.n_p = L.n, self.B.shape[0]
        # state z = (u, p): zero with the Dirichlet values imposed (what Firedrake does to the initial guess).  With the
        # device-side refresh the state LIVES on the device (``_dz``); ``u`` / ``p`` fetch it when somebody asks.
        self._host_u = np.zeros(self.n_u)
        bc_nodes = L.V.bc_nodes
        self._host_u.reshape(-1, dim)[bc_nodes] = problem.driver(L.V.node_coords[bc_nodes])
        self._host_p = np.zeros(self.n_p)
        self._device_newer = False          # the device copy of the state is ahead of the host arrays
        self._device_current = False        # the device copy equals the host arrays
        self._load = None
        self.area = float(self.vol.sum())

    # -- the state: host arrays on request, resident on the device during the solves -------------------------------------
    def _fetch_state(self):
        if self._device_newer:
            z = self._dz.get()
            self._host_u, self._host_p = z[:self.n_u].copy(), z[self.n_u:].copy()
            self._device_newer, self._device_current = False, True

    @property
    def u(self):
        self._fetch_state()
        return self._host_u

    @u.setter
    def u(self, value):
        self._fetch_state()
        self._host_u = np.asarray(value, dtype=np.float64)
        self._device_current = False

    @property
    def p(self):
        self._fetch_state()
        return self._host_p

    @p.setter
    def p(self, value):
        self._fetch_state()
        self._host_p = np.asarray(value, dtype=np.float64)
        self._device_current = False

    # -- device side (overridden by alfi_amd.dist.DistNavierStokesSolver for partitioned levels) -------------------------
    def _any_rank(self, flag):
        """True if ``flag`` holds on any rank (one rank here)."""
        return bool(flag)

    def _lazy_generation(self):
        """Operators and transfers as recipes that assemble the rows somebody asks for (alfi_amd.lazy) instead of global
        values: for the partitioned solver, whose ranks only ever need their own rows."""
        return False

    def _create_device(self, restriction):
        self.ctx = self._ctx_arg or hip.Context(0)
        self.hmg = HipMG(self.ctx, self.levels, self.transfers, self.params["fieldsplit_0"], restriction=restriction)
        self.saddle = hip.Saddle(self.hmg.mg, self.B, None if self.sv else self.vol, self.nu, self.gamma,
                                 remove_constant_nullspace=self.nullspace, mass_inv=self.Minv)

    def _push_operators(self):
        """New operator values on every level: re-gather and re-invert the patches, new coarse inverse."""
        self.hmg.update(self.levels)
        self.hmg.mg.levels[0].update_values(self.levels[0].A.vals)
        self.hmg.mg.levels[0].coarse_factor_auto()

    # -- operator refresh on the device ---------------------------------------------------------------------------------------
    def _device_assembly_possible(self):
        return True

    def _setup_device_assembly(self):
        """Once: the cells, the reference tensors of the element and the contributor lists of every level go to the device
        (alfi_level_set_assembly: every term of the operator is formed there, cell by cell); per-level state vectors for the
        injected field."""
        self._dstate = []
        for L, dl in zip(self.levels, self.hmg.mg.levels):
            dl.set_assembly(L.V, L.A.rowptr, L.A.colidx, full_div=self.sv)
            if self.supg:
                dl.set_supg(L.V)
            self._dstate.append(self.ctx.vec(L.n))
        # the Newton state z = (u | p), the residual F and the update live on the device; the finest level's state vector IS
        # the velocity part of z
        n = self.levels[-1].n + self.B_raw.shape[0]
        self._dz, self._dF, self._dd = self.ctx.vec(n), self.ctx.vec(n), self.ctx.vec(n)
        self._dstate[-1] = hip.view(self._dz, 0, self.levels[-1].n)
        self._dres = self.ctx.vec(self.levels[-1].n)
        # the residual's divergence products on the device too (B with ALL columns; the Jacobian's B lives in the saddle
        # solver): at config-4 size the two host products were 0.13 s of a 0.2 s residual
        self._dB = hip.Csr(self.ctx, self.B_raw)
        self._dBT = hip.Csr(self.ctx, self.B_raw.T.tocsr())
        self._dp, self._dFp = self.ctx.vec(self.B_raw.shape[0]), self.ctx.vec(self.B_raw.shape[0])
        self._asm_ready = True

    def _first_operators_on_device(self):
        """The Stokes operators the hierarchy is created with (what build_hierarchy(Re=0) assembles on the host otherwise),
        formed on the device; patches and coarse grid factored."""
        mgl = self.hmg.mg.levels
        for dl in mgl:
            dl.assemble(self.nu, self.gamma, 0.0, None, True)
        for L, dl in zip(self.levels, mgl):
            if L.level > 0:
                dl.factor_with_fallback()
        mgl[0].coarse_factor_auto()
        self.ctx.sync()

    def _device_states(self, u):
        """Current velocity on every level, on the device: the finest is the velocity part of the resident state (``u`` given:
        uploaded into it first), the coarser ones by alfi_inject (solver.py:595) -- an index map on the nested hierarchies, the
        point-evaluation matrix sv.bary_injection on the barycentric ones (the third entry of the reference's transfer triple,
        solver.py:645-652)."""
        if u is not None:
            self._fetch_state()
            self._dstate[-1].set(u)
            self._device_current = False
        for l in range(len(self.levels) - 1, 0, -1):
            self.hmg.mg.transfers[l - 1].inject(self._dstate[l], self._dstate[l - 1])

    def _rediscretise_device(self, u, adv):
        t0 = time.time()
        self._device_states(u)
        mgl = self.hmg.mg.levels
        for dl, st in zip(mgl, self._dstate):
            if adv and self.supg:     # A = nu K + gamma D + N(w) + the linearised SUPG term, THEN the boundary conditions
                dl.assemble_supg(self.nu, self.gamma, adv, st, self.supg_weight, self.supg_magic, True)
            else:
                dl.assemble(self.nu, self.gamma, adv, st if adv else None, True)
        self.ctx.sync()
        t1 = time.time()
        for L, dl in zip(self.levels, mgl):
            L.nu = self.nu
            if L.level > 0:
                dl.factor_with_fallback()
        mgl[0].coarse_factor_auto()
        self.ctx.sync()
        self.timings["assemble_s"] += t1 - t0
        self.timings["factor_s"] += time.time() - t1

    def _residual_device(self, u, p, adv):
        """F_u = (nu K + gamma D) u + 1/2 N(u) u + B^T p - f: one matrix-free product, cell by cell, with HALF the advection
        term and without boundary conditions (N(u) u = 2 (u . grad) u)."""
        self._fetch_state()                                                # (the host arrays keep what the device held)
        self._dz.set(np.concatenate([u, p]))
        self._device_current = False
        self._residual_on_device(adv)
        F = self._dF.get()
        return F[:self.n_u], F[self.n_u:]

    def _residual_on_device(self, adv):
        """F(z) for the state resident in ``_dz`` into ``_dF`` = (F_u | F_p); nothing crosses to the host."""
        fin = self.hmg.mg.levels[-1]
        n_u, n_p = self.n_u, self.n_p
        du, dp = self._dstate[-1], hip.view(self._dz, n_u, n_p)
        Fu, Fp = hip.view(self._dF, 0, n_u), hip.view(self._dF, n_u, n_p)
        fin.assemble_mult(self.nu, self.gamma, 0.5 * adv, du if adv else None, du, Fu)
        if adv and self.supg:         # + the SUPG residual, gathered on the device into the same vector
            fin.supg(self.nu, self.supg_weight, self.supg_magic, du, False, Fu)
        self._dBT.mult(dp, Fu, mode=2)                                    # F_u += B^T p
        if self._load is not None:                                         # body force: F_u -= (f, v)
            self.ctx.axpy(Fu, self._dload, -1.0)
        fin.zero_bc(Fu)                                                    # bc.zero(F), solver.py:282-286
        self._dB.mult(du, Fp)                                              # F_p = B u

    def _set_parameters(self):
        for T, dt in zip(self.transfers, self.hmg.mg.transfers):            # AutoSchoeberlTransfer.rebuild, transfer.py:173-184
            if T.nu != self.nu:
                T.nu = self.nu
                dt.update(self.nu, self.gamma)
        self.saddle.update(self.nu, self.gamma)

    def _solve_on_device(self, re, adv):
        """The Newton loop of ``solve`` with the state, the residual and the update resident in HBM: per step the operators
        are refreshed from the state in place, J e = F is solved on the device (e = - update; the Krylov iterates of b and
        - b mirror each other), z -= e, and only scalars -- norms, iteration counts -- reach the host."""
        if not (self._device_newer or self._device_current):
            self._push_state()
            self._device_current = True
        if self._load is not None:
            self._push_load()
        norm = lambda v: float(np.sqrt(self._zdot(v, v)))
        lin_its, newton_its = 0, 0
        t_r = time.time()
        self._residual_on_device(adv)
        f0 = fnorm = norm(self._dF)
        self.timings["residual_s"] += time.time() - t_r
        hist = [fnorm]
        small_step = False
        while fnorm > max(self.snes_rtol * f0, self.snes_atol) and newton_its < self.snes_max_it and not small_step:
            self._rediscretise_device(None, adv)
            t_s = time.time()
            its, rn = self._zsolve(self._dF, self._dd)
            self._zaxpy(self._dz, self._dd, -1.0)
            self._device_newer, self._device_current = True, False
            self.timings["solve_s"] += time.time() - t_s
            lin_its += its
            newton_its += 1
            self.timings["newton_steps"] += 1
            t_r = time.time()
            self._residual_on_device(adv)
            fnorm = norm(self._dF)
            self.timings["residual_s"] += time.time() - t_r
            hist.append(fnorm)
            # SNESConvergedDefault [3P] with snes_stol (solver.py:490, 498): the step is small relative to the iterate
            if norm(self._dd) < self.snes_stol * norm(self._dz):
                small_step = True
            if self.verbose:
                print("[alfi_amd] Re %g  Newton %d  |F| %.3e  (%d Krylov its, linear residual %.2e)"
                      % (re, newton_its, fnorm, its, rn), flush=True)
        if self.nullspace:                                                   # zero pressure integral, solver.py:273-277
            self._shift_pressure()
            self._device_newer, self._device_current = True, False
        return lin_its, newton_its, hist, small_step, fnorm, f0

    # -- the pieces of the device-resident loop a partitioned solver replaces (alfi_amd.dist.DistNavierStokesSolver) ------
    def _push_state(self):
        self._dz.set(np.concatenate([self._host_u, self._host_p]))

    def _push_load(self):
        if getattr(self, "_dload", None) is None:
            self._dload = self.ctx.vec(self.n_u)
        self._dload.set(self._load)

    def _zdot(self, x, y):
        return self.saddle.dot(x, y)

    def _zsolve(self, b, x):
        return self.saddle.solve(b, x, self.rtol, self.atol, self.params["ksp_max_it"], 30)

    def _zaxpy(self, y, x, a):
        self.ctx.axpy(y, x, a)

    def _shift_pressure(self):
        """p -= (int p) / |domain| on the device."""
        ctx = self.ctx
        if getattr(self, "_dvolz", None) is None:
            self._dvolz = ctx.vec(np.concatenate([np.zeros(self.n_u), self.vol]))
            self._dones = ctx.vec(np.ones(self.n_p))
        ctx.axpy(self._dz, self._dones, -self._zdot(self._dvolz, self._dz) / self.area, n=self.n_p, y_off=self.n_u)

    def _device_state_resident(self):
        """Whether the Newton loop runs with the state on the device (the operators are refreshed there anyway)."""
        return self.device_assembly

    def _linear_solve(self, rhs):
        """J d = rhs for the current operators: (d, Krylov iterations, true residual norm)."""
        db, dx = self.ctx.vec(rhs), self.ctx.vec(self.n_u + self.n_p)
        its, rn = self.saddle.solve(db, dx, self.rtol, self.atol, self.params["ksp_max_it"], 30)
        return dx.get(), its, rn

    def close(self):
        if getattr(self, "_asm_ready", False):
            self._dB.close()
            self._dBT.close()
        self.saddle.close()
        self.hmg.mg.close()

    # -- host side: state on all levels, operators, residual -------------------------------------------------------------
    def _winds(self, u):
        """Current velocity as nodal field on every level (finest given, coarser by inject, solver.py:595)."""
        d = self.problem.dim
        w = [None] * len(self.levels)
        w[-1] = u.reshape(-1, d)
        for l in range(len(self.levels) - 1, 0, -1):
            T = self.transfers[l - 1]
            # nested uniform hierarchy: every coarse node is a fine node; bary hierarchy: point evaluation (sv.bary_injection)
            w[l - 1] = w[l][T.inject_map] if T.inject_map is not None else T.inject_matrix @ w[l]
        return w

    def level_values(self, L, state, adv, with_bc):
        """BSR values of the linearised momentum block of level L about ``state`` (num_nodes, dim): viscous + grad-div +
        Newton-linearised advection (+ the linearised SUPG term, ``advect * stabilisation_form``, solver.py:233-234)."""
        state = np.ascontiguousarray(state)
        A = _assemble(L, self.nu, self.gamma, adv, state, False, self.sv)
        if adv and self.supg:
            _hostlib.supg(L.V, state, self.nu, self.supg_weight, self.supg_magic, L.A.rowptr, L.A.colidx, A)
        if with_bc:
            _hostlib.apply_bc_bsr(L.V.num_nodes, L.V.dim, L.A.rowptr, L.A.colidx, A, np.repeat(L.V.bc_node_mask, L.V.dim))
        return A

    def _rediscretise(self, u, adv):
        if self.device_assembly:
            return self._rediscretise_device(u, adv)
        winds = self._winds(u)
        for L, w in zip(self.levels, winds):
            L.A = BSR(L.A.nbrows, L.A.nbcols, L.bs, L.A.rowptr, L.A.colidx, self.level_values(L, w, adv, True))
            L.nu = self.nu
        self._push_operators()

    def residual(self, u, p, adv):
        """F(u, p) of solver.py:565-568 (rhs = 0), Dirichlet rows zeroed (``bc.zero(F)``, solver.py:282-286)."""
        if self.device_assembly:
            return self._residual_device(u, p, adv)
        L = self.levels[-1]
        d = self.problem.dim
        wind = np.ascontiguousarray(u.reshape(-1, d))
        A0 = BSR(L.A.nbrows, L.A.nbcols, L.bs, L.A.rowptr, L.A.colidx,
                 _assemble(L, self.nu, self.gamma, 0.0, None, False, self.sv)).to_scipy()
        Fu = A0 @ u
        if adv:
            J = BSR(L.A.nbrows, L.A.nbcols, L.bs, L.A.rowptr, L.A.colidx,
                    _assemble(L, self.nu, self.gamma, 1.0, wind, False, self.sv)).to_scipy()
            Fu = 0.5 * (Fu + J @ u)                  # A0 u + 1/2 N(u) u with N = J - A0
            if self.supg:
                Fs = np.zeros_like(Fu)
                _hostlib.supg(L.V, wind, self.nu, self.supg_weight, self.supg_magic, F=Fs)
                Fu = Fu + Fs
        Fu = Fu + self.B_raw.T @ p
        if self._load is not None:                   # body force (manufactured solutions, examples/mms.py): F_u -= (f, v)
            Fu = Fu - self._load
        Fu[L.bc_dofs] = 0.0
        Fp = self.B_raw @ u
        return Fu, Fp

    # -- the solve loop ---------------------------------------------------------------------------------------------------
    def solve(self, re):
        t0 = time.time()
        if re == 0:
            adv, self.nu = 0.0, self.char_L * self.char_U                   # Stokes, solver.py:261-264
        else:
            adv, self.nu = 1.0, self.char_L * self.char_U / re
        self._set_parameters()
        self._load = None
        if hasattr(self.problem, "rhs"):             # NavierStokesProblem.rhs (problem.py:46-47 of the reference): default none
            if self.supg:
                raise NotImplementedError("a body force with SUPG: the stabilisation's strong residual (stabilisation.py:"
                                          "86-91) would have to carry it; built for rhs = 0 only")
            from .mms import load_vector
            self._load = load_vector(self.levels[-1].V, lambda x: self.problem.rhs(x, re))
        if self._device_state_resident():
            lin_its, newton_its, hist, small_step, fnorm, f0 = self._solve_on_device(re, adv)
            info = {"Re": re, "nu": self.nu, "linear_iter": lin_its, "nonlinear_iter": newton_its,
                    "time": (time.time() - t0) / 60.0, "residual_history": hist,
                    "converged": small_step or fnorm <= max(self.snes_rtol * f0, self.snes_atol),
                    "converged_reason": "SNORM_RELATIVE" if small_step else "FNORM"}
            return _StateHandle(self), info
        u, p = self.u.copy(), self.p.copy()
        lin_its, newton_its = 0, 0
        t_r = time.time()
        Fu, Fp = self.residual(u, p, adv)
        self.timings["residual_s"] += time.time() - t_r
        f0 = fnorm = float(np.sqrt(Fu @ Fu + Fp @ Fp))
        hist = [fnorm]
        small_step = False
        while fnorm > max(self.snes_rtol * f0, self.snes_atol) and newton_its < self.snes_max_it and not small_step:
            self._rediscretise(u, adv)
            rhs = -np.concatenate([Fu, Fp])
            t_s = time.time()
            delta, its, rn = self._linear_solve(rhs)
            self.timings["solve_s"] += time.time() - t_s
            u += delta[:self.n_u]
            p += delta[self.n_u:]
            lin_its += its
            newton_its += 1
            self.timings["newton_steps"] += 1
            t_r = time.time()
            Fu, Fp = self.residual(u, p, adv)
            self.timings["residual_s"] += time.time() - t_r
            fnorm = float(np.sqrt(Fu @ Fu + Fp @ Fp))
            hist.append(fnorm)
            # SNESConvergedDefault [3P] with snes_stol (solver.py:490, 498): the step is small relative to the iterate
            if np.linalg.norm(delta) < self.snes_stol * float(np.sqrt(u @ u + p @ p)):
                small_step = True
            if self.verbose:
                print("[alfi_amd] Re %g  Newton %d  |F| %.3e  (%d Krylov its, linear residual %.2e)"
                      % (re, newton_its, fnorm, its, rn), flush=True)
        if self.nullspace:
            p -= (self.vol @ p) / self.area                                  # zero pressure integral, solver.py:273-277
        self.u, self.p = u, p
        info = {"Re": re, "nu": self.nu, "linear_iter": lin_its, "nonlinear_iter": newton_its,
                "time": (time.time() - t0) / 60.0, "residual_history": hist,
                "converged": small_step or fnorm <= max(self.snes_rtol * f0, self.snes_atol),
                "converged_reason": "SNORM_RELATIVE" if small_step else "FNORM"}
        return (u, p), info


class _StateHandle(object):
    """What ``solve`` returns for the state while it lives on the device (the reference returns the Function z, a handle as
    well): unpacking it -- ``u, p = z`` -- fetches the arrays."""

    def __init__(self, solver):
        self._s = solver

    def __iter__(self):
        return iter((self._s.u, self._s.p))


def run_solver(solver, res):
    """alfi.driver.run_solver (driver.py:95-128) without checkpoints / ParaView output: continuation in Re."""
    results = {}
    for re in res:
        _, results[re] = solver.solve(re)
    return results


# PETSc event names the reference's report prints (driver.py:80) -> the library's profiling classes
from ._lib import PETSC_EVENT_NAMES as _PETSC
_EVENT_NAMES = [(_PETSC[k], k) for k in ("PATCH_APPLY", "PATCH_SCATTER", "PATCH_FACTOR", "MATMULT", "BLAS1", "PROLONG", "RESTRICT",
                                         "COARSE")]


def performance_info(solver, out=print):
    """alfi.driver.performance_info (driver.py:77-92): device time per event class since profiling was switched on
    (``solver.ctx.prof_enable(True)`` before the solves), sorted, with time per 1k dofs."""
    prof = solver.ctx.prof_get()
    ndofs = solver.n_u + solver.n_p
    rows = sorted(((name, prof[key][0] * 1e-3, prof[key][1]) for name, key in _EVENT_NAMES), key=lambda r: -r[1])
    out("Some performance info:")
    for name, t, cnt in rows:
        out(("%s:" % name).ljust(30) + "Time = % 6.2fs, Time/1kdofs = %.2fs  (%d launches)" % (t, 1000 * t / ndofs, cnt))
    return rows
