"""Front end with the reference's plug-in surface for the velocity-block multigrid (alfi/solver.py).

* ``HipPatchPC``: a PCPython-protocol class (``initialize / update / apply / applyTranspose``; the in-tree example of
  that protocol is ``DGMassInv``, solver.py:15-38) that is selected exactly like the reference selects
  ``firedrake.PatchPC``: ``"pc_type": "python", "pc_python_type": "alfi_amd.HipPatchPC"`` (solver.py:318-319).  It reads
  the ``patch_pc_patch_*`` / ``patch_sub_*`` keys the reference sets (solver.py:320-344, 599-602) and runs PCPATCH's
  setup and apply on the GPU.
* ``mg_levels_solver`` / ``fieldsplit_0_mg``: the option dictionaries of ``get_parameters`` (solver.py:313-344, 359-379)
  with the PatchPC swapped for ``HipPatchPC``.
* ``HipMG``: drives PCMG (full or multiplicative V) + FGMRES(k) from such a dictionary -- the stand-in for PETSc when
  petsc4py is not importable (it is not, in this image).  One ``apply`` = what ``fieldsplit_0``'s Richardson(1)/PCMG
  does to a right-hand side.
"""
import importlib

import numpy as np

from . import hip
from .relaxation import Options, PlexLike, patch_points_to_dofs

SUPPORTED_PATCH_KEYS = {
    "patch_pc_patch_save_operators", "patch_pc_patch_partition_of_unity", "patch_pc_patch_local_type",
    "patch_pc_patch_statistics", "patch_pc_patch_symmetrise_sweep", "patch_pc_patch_precompute_element_tensors",
    "patch_pc_patch_construct_type", "patch_pc_patch_construct_dim", "patch_pc_patch_construct_python_type",
    "patch_pc_patch_sub_mat_type", "patch_pc_patch_dense_inverse", "patch_pc_patch_multiplicative",
    "patch_sub_ksp_type", "patch_sub_pc_type", "patch_sub_pc_factor_mat_solver_type",
}


def _truthy(v):
    return v if isinstance(v, bool) else str(v).lower() in ("1", "true", "yes")


def mg_levels_solver(tdim, patch="star", patch_composition="additive", smoothing=None, relaxation_direction=None):
    """The ``mg_levels_solver`` dictionary of alfi/solver.py:313-344 with ``configure_patch_solver`` of
    ConstantPressureSolver (solver.py:599-602) applied."""
    multiplicative = patch_composition == "multiplicative"
    if multiplicative and relaxation_direction is None:
        raise NotImplementedError("Need to specify a relaxation_direction in the problem.")
    if smoothing is None:
        smoothing = 10 if tdim > 2 else 6
    opts = {
        "ksp_type": "fgmres",
        "ksp_norm_type": "unpreconditioned",
        "ksp_max_it": smoothing,
        "ksp_convergence_test": "skip",
        "pc_type": "python",
        "pc_python_type": "alfi_amd.HipPatchPC",
        "patch_pc_patch_save_operators": True,
        "patch_pc_patch_partition_of_unity": False,
        "patch_pc_patch_local_type": "multiplicative" if multiplicative else "additive",
        "patch_pc_patch_statistics": False,
        "patch_pc_patch_symmetrise_sweep": multiplicative,
        "patch_pc_patch_precompute_element_tensors": True,
        "patch_sub_ksp_type": "preonly",
        "patch_sub_pc_type": "lu",
        "patch_pc_patch_sub_mat_type": "seqdense",
        "patch_sub_pc_factor_mat_solver_type": "petsc",
        "patch_pc_patch_dense_inverse": True,
    }
    if patch == "star":
        if multiplicative:
            opts["patch_pc_patch_construct_type"] = "python"
            opts["patch_pc_patch_construct_python_type"] = "alfi_amd.Star"
            opts["patch_pc_patch_construction_Star_sort_order"] = relaxation_direction
        else:
            opts["patch_pc_patch_construct_type"] = "star"
            opts["patch_pc_patch_construct_dim"] = 0
    elif patch == "macro":
        opts["patch_pc_patch_construct_type"] = "python"
        opts["patch_pc_patch_construct_python_type"] = "alfi_amd.MacroStar"
        opts["patch_pc_patch_construction_MacroStar_sort_order"] = relaxation_direction
    else:
        raise NotImplementedError("Unknown patch type %s" % patch)
    return opts


def configure_patch_solver_sv(opts, tdim, use_mkl=False):
    """ScottVogeliusSolver.configure_patch_solver (solver.py:655-659)."""
    opts = dict(opts)
    opts["patch_pc_patch_sub_mat_type"] = "seqaij"
    opts["patch_sub_pc_factor_mat_solver_type"] = ("mkl_pardiso" if use_mkl else "umfpack") if tdim > 2 else "petsc"
    opts.pop("patch_pc_patch_dense_inverse", None)
    return opts


def fieldsplit_0_mg(mg_levels):
    """alfi/solver.py:359-379 (the coarse solve is a dense inverse applied on the GPU instead of telescoped
    SuperLU_DIST; the keys are accepted and ignored)."""
    return {
        "ksp_type": "richardson",
        "ksp_richardson_self_scale": False,
        "ksp_max_it": 1,
        "ksp_norm_type": "unpreconditioned",
        "ksp_convergence_test": "skip",
        "pc_type": "mg",
        "pc_mg_type": "full",
        "pc_mg_log": None,
        "mg_levels": mg_levels,
        "mg_coarse_pc_type": "python",
        "mg_coarse_pc_python_type": "firedrake.AssembledPC",
        "mg_coarse_assembled": {"mat_type": "aij", "pc_type": "lu"},
    }


def _resolve(dotted):
    mod, _, name = dotted.rpartition(".")
    return getattr(importlib.import_module(mod), name)


def macro_vertex_labels(mesh):
    """``MacroVertices`` label (alfi/bary.py:18-19 sets it to 1 on the vertices of the mesh that is then split).  For a
    uniformly refined mesh the analogous macro vertices are the vertices inherited from the parent mesh; a mesh without
    parent information carries no label."""
    if getattr(mesh, "macro_vertex_mask", None) is not None:          # Alfeld split (mesh.bary_refine): the split mesh's vertices
        return {"MacroVertices": {int(mesh.num_cells + v): 1 for v in np.flatnonzero(mesh.macro_vertex_mask)}}
    if getattr(mesh, "vertex_parents", None) is None:
        return {}
    vp_ = np.asarray(mesh.vertex_parents)
    macro = np.flatnonzero(vp_[:, 0] == vp_[:, 1])
    nc = mesh.num_cells
    return {"MacroVertices": {int(nc + v): 1 for v in macro}}


class PC(object):
    """The small part of a PETSc PC object a PCPython class touches: operators, DM, options prefix (+ attributes)."""

    def __init__(self, ctx, level_data, options=None, prefix=""):
        self.ctx = ctx                  # alfi_amd.hip.Context
        self.level_data = level_data    # alfi_amd.problem.LevelData (operator, space, Dirichlet dofs)
        self.options = dict(options or {})
        self.prefix = prefix
        self._dm = PlexLike(level_data.V.mesh, labels=macro_vertex_labels(level_data.V.mesh))
        self.attrs = {}

    def getOperators(self):
        return self.level_data.A, self.level_data.A

    def getDM(self):
        return self._dm

    def getOptionsPrefix(self):
        return self.prefix

    def getAttr(self, k):
        return self.attrs.get(k)

    def setAttr(self, k, v):
        self.attrs[k] = v


def _as_device(ctx, v, n):
    """(DeviceVec, writeback) for a DeviceVec, a NumPy array, or anything with getArray() (petsc4py Vec)."""
    if isinstance(v, hip.DeviceVec):
        return v, None
    arr = v.getArray() if hasattr(v, "getArray") else v
    arr = np.asarray(arr)
    assert arr.shape == (n,)
    return ctx.vec(arr), arr


class HipPatchPC(object):
    """PCPATCH on the GPU behind the PCPython protocol (solver.py:15-38 shows the protocol)."""

    def initialize(self, pc):
        opts = Options(pc.getOptionsPrefix(), pc.options)
        L = pc.level_data
        unknown = [k for k in pc.options if k.startswith(pc.getOptionsPrefix() + "patch_")
                   and k[len(pc.getOptionsPrefix()):] not in SUPPORTED_PATCH_KEYS
                   and "patch_pc_patch_construction_" not in k]
        if unknown:
            raise ValueError("unsupported PatchPC options: %s" % unknown)
        local_type = opts.getString("patch_pc_patch_local_type", "additive")
        if _truthy(opts.getString("patch_pc_patch_multiplicative", "false")):
            local_type = "multiplicative"
        if local_type not in ("additive", "multiplicative"):
            raise NotImplementedError("patch local_type %r" % local_type)
        self.multiplicative = local_type == "multiplicative"
        self.symmetrise = _truthy(opts.getString("patch_pc_patch_symmetrise_sweep", "false"))
        self.partition_of_unity = _truthy(opts.getString("patch_pc_patch_partition_of_unity", "false"))
        if self.partition_of_unity and self.multiplicative:
            raise NotImplementedError("partition_of_unity with multiplicative sweeps")
        sub_mat = opts.getString("patch_pc_patch_sub_mat_type", "seqdense")
        if sub_mat not in ("seqdense", "dense", "seqaij", "aij"):
            raise NotImplementedError("patch sub_mat_type %r" % sub_mat)
        # seqaij (ScottVogeliusSolver.configure_patch_solver, solver.py:655-659: sparse patch matrices factored by
        # UMFPACK / PARDISO) asks for the same operator A_p^-1 as seqdense + dense_inverse (solver.py:599-602); here both
        # are the explicit dense inverse -- macro-star patches are inverted on the FP64 matrix cores
        ctype = opts.getString("patch_pc_patch_construct_type", "star")
        cdim = opts.getInt("patch_pc_patch_construct_dim", 0)
        ctor = None
        if ctype == "star" and cdim == 0:
            ptr, dofs, _ = L.V.star_patches()
            self.iterset = np.arange(len(ptr) - 1)
        elif ctype in ("star", "python"):
            if ctype == "star":
                # built-in stars of edges / faces / cells (patch_pc_patch_construct_dim != 0; the reference passes 0,
                # solver.py:338): the same point sets as the python Star constructor seeded on that stratum
                from .relaxation import Star
                ctor = Star()
                pc.options = dict(pc.options)
                pc.options[pc.getOptionsPrefix() + "patch_pc_patch_construction_Star_dim"] = cdim
            else:
                ctor = _resolve(opts.getString("patch_pc_patch_construct_python_type"))()
            # firedrake.PatchPC hands the constructor its inner PCPATCH object, whose options prefix is the outer one
            # + "patch_" [3P]: that is where pc_patch_construction_<Name>_sort_order lives (solver.py:335, 342)
            inner = PC(pc.ctx, pc.level_data, options=pc.options, prefix=pc.getOptionsPrefix() + "patch_")
            inner._dm = pc.getDM()
            patches, iterset = ctor(inner)
            ptr, dofs, kept = patch_points_to_dofs(L.V, pc.getDM(), patches)
            # the iteration set indexes the constructor's patch list; patches without free dofs were dropped (PCPATCH
            # skips them in the sweep: `if (len <= 0) continue` [3P])
            new = np.full(len(patches), -1, dtype=np.int64)
            new[np.asarray(kept, dtype=np.int64)] = np.arange(len(kept))
            it = new[np.asarray(iterset, dtype=np.int64)]
            self.iterset = it[it >= 0]
        else:
            raise NotImplementedError("patch construct_type %r" % ctype)
        self.patch_ptr, self.patch_dofs = ptr, dofs
        self.level = hip.Level(pc.ctx, L.A, L.bc_dofs)
        self.level.set_patches(ptr, dofs)
        # macro-star patches on an Alfeld-split mesh (ScottVogeliusSolver: the reference keeps SPARSE patch factors there,
        # solver.py:655-659): store the factors condensed -- interiors of the macro cells + skeleton -- unless the sweep is
        # multiplicative (that kernel multiplies with dense inverses)
        self.condensed = False
        import os
        if (ctype == "python" and getattr(L.V.mesh, "macro_mesh", None) is not None and not self.multiplicative
                and os.environ.get("ALFI_CONDENSE", "1") != "0" and type(ctor).__name__ == "MacroStar"
                and ctype == "python"):
            from .sv import macro_cell_groups
            self.level.set_patch_groups(macro_cell_groups(L.V, dofs))
            self.condensed = True
        # (operator values not there yet -- formed on the device by the caller, who then factors: problem.build_hierarchy
        # with operator_values=False)
        if L.A.vals is not None:
            self.level.factor()
        if self.partition_of_unity:
            self.level.set_partition_of_unity(True)
        self.wavefronts = self.level.set_multiplicative(self.iterset, self.symmetrise) if self.multiplicative else 0
        self.n = L.n

    def update(self, pc):
        """New operator values (Newton step / Reynolds continuation): re-gather and re-invert every patch, which is what
        PatchPC.update -> PCSetUp_PATCH does with save_operators (solver.py:320)."""
        self.level.update_values(pc.level_data.A.vals)
        self.level.factor()

    def apply(self, pc, x, y):
        dx, _ = _as_device(pc.ctx, x, self.n)
        dy, ywb = _as_device(pc.ctx, y, self.n)
        self.level.patch_apply(dx, dy)
        if ywb is not None:
            ywb[:] = dy.get()

    def applyTranspose(self, pc, x, y):
        raise NotImplementedError("Sorry!")


class HipMG(object):
    """PCMG + KSPFGMRES(k) driven by the reference's option dictionary (solver.py:359-379), device resident."""

    def __init__(self, ctx, levels, transfers, params, restriction=False, coarse_inv=None):
        if params.get("pc_type") != "mg":
            raise ValueError("expected the fieldsplit_0_mg dictionary (pc_type mg)")
        mgl = params["mg_levels"]
        if mgl.get("ksp_type") != "fgmres" or mgl.get("ksp_convergence_test") != "skip":
            raise NotImplementedError("level smoother must be fgmres with convergence_test skip (solver.py:314-317)")
        if mgl.get("pc_type") != "python":
            raise NotImplementedError("level pc_type must be python")
        self.k = int(mgl["ksp_max_it"])
        self.full = params.get("pc_mg_type", "multiplicative") == "full"
        pc_cls = _resolve(mgl["pc_python_type"])
        self.ctx = ctx
        self.pcs, self.pc_objs = [], []
        dlevels = []
        for L in levels:
            if L.level == 0:
                dl = hip.Level(ctx, L.A, L.bc_dofs)
                if coarse_inv is not None:
                    dl.set_coarse_inverse(coarse_inv)
                elif L.A.vals is None:      # values formed on the device later: remember the choice, factor then
                    dl._coarse_choice = ("auto", None)
                else:
                    dl.coarse_factor_auto(getattr(getattr(L, "V", None), "node_coords", None))
                dlevels.append(dl)
                self.pcs.append(None)
                self.pc_objs.append(None)
                continue
            pc = PC(ctx, L, options=mgl)
            obj = pc_cls()
            obj.initialize(pc)
            self.pcs.append(pc)
            self.pc_objs.append(obj)
            dlevels.append(obj.level)
        self.mg = hip.Multigrid.__new__(hip.Multigrid)
        hip.Multigrid._from_device_levels(self.mg, ctx, dlevels, transfers, self.k, restriction)
        self.n = levels[-1].n

    def update(self, levels):
        for pc, obj, L in zip(self.pcs, self.pc_objs, levels):
            if obj is not None:
                pc.level_data = L
                obj.update(pc)

    def apply(self, b, x):
        """x <- PCMG(b): one full cycle (pc_mg_type full) or one V-cycle from a zero initial guess."""
        db, _ = _as_device(self.ctx, b, self.n)
        dx, xwb = _as_device(self.ctx, x, self.n)
        if self.full:
            self.mg.fcycle(db, dx)
        else:
            dx.zero()
            self.mg.vcycle(db, dx)
        if xwb is not None:
            xwb[:] = dx.get()


class DGMassInv(object):
    """The pressure-block preconditioner of the reference (alfi/solver.py:15-38), same PCPython protocol:
    ``y = -(nu + gamma) M_p^-1 x`` with the (diagonal, P0) pressure mass matrix.  ``pc.getAttr`` style context: the
    ``PC`` passed in must carry ``attrs["nu"], attrs["gamma"], attrs["mass_diag"]`` (the reference pulls nu and gamma
    from the application context, solver.py:17-21)."""

    def initialize(self, pc):
        self.update(pc)

    def update(self, pc):
        self.nu, self.gamma = float(pc.getAttr("nu")), float(pc.getAttr("gamma"))
        self.minv = 1.0 / np.asarray(pc.getAttr("mass_diag"), dtype=np.float64)

    def apply(self, pc, x, y):
        y[...] = -(self.nu + self.gamma) * self.minv * np.asarray(x)

    def applyTranspose(self, pc, x, y):
        raise NotImplementedError("Sorry!")


def outer_solver(tdim, fieldsplit_0, high_accuracy=False):
    """The ``outer`` dictionary of alfi/solver.py:402-499 for solver_type almg (Newton keys included verbatim; only the
    linear-solve keys are acted on by ``HipOuterSolver``)."""
    outer = {
        "snes_type": "newtonls", "snes_max_it": 20, "snes_linesearch_type": "basic", "snes_linesearch_maxstep": 1.0,
        "snes_monitor": None, "snes_linesearch_monitor": None, "snes_converged_reason": None,
        "ksp_type": "fgmres", "ksp_monitor_true_residual": None, "ksp_converged_reason": None,
        "mat_type": "nest", "ksp_max_it": 500,
        "pc_type": "fieldsplit", "pc_fieldsplit_type": "schur",
        "pc_fieldsplit_schur_factorization_type": "full", "pc_fieldsplit_schur_precondition": "user",
        "fieldsplit_0": fieldsplit_0,
        "fieldsplit_1": {"ksp_type": "preonly", "pc_type": "python", "pc_python_type": "alfi_amd.solver.DGMassInv"},
    }
    if high_accuracy:
        outer.update({"ksp_rtol": 1.0e-12, "ksp_atol": 1.0e-12})
    elif tdim == 2:
        outer.update({"ksp_rtol": 1.0e-9, "ksp_atol": 1.0e-10})
    else:
        outer.update({"ksp_rtol": 1.0e-8, "ksp_atol": 1.0e-8})
    return outer


class HipOuterSolver(object):
    """One linear solve of the reference's outer iteration (solver.py:386-422) on the GPU: KSPFGMRES (restart 30, the
    PETSc default the reference does not override) around PCFIELDSPLIT-Schur-full, fieldsplit_0 = the device PCMG
    (``HipMG``), fieldsplit_1 = ``DGMassInv``.  ``solve(f, g)`` returns (u, p, iterations, true residual norm)."""

    def __init__(self, ctx, levels, transfers, params, restriction=False, coarse_inv=None):
        from .problem import build_pressure_coupling
        if params.get("ksp_type") != "fgmres" or params.get("pc_type") != "fieldsplit" \
                or params.get("pc_fieldsplit_type") != "schur" \
                or params.get("pc_fieldsplit_schur_factorization_type") != "full" \
                or params.get("pc_fieldsplit_schur_precondition") != "user":
            raise NotImplementedError("outer solver must be fgmres + fieldsplit schur/full/user (solver.py:402-411)")
        fs1 = params.get("fieldsplit_1", {})
        if fs1.get("ksp_type") != "preonly" or not str(fs1.get("pc_python_type", "")).endswith("DGMassInv"):
            raise NotImplementedError("fieldsplit_1 must be preonly + DGMassInv (solver.py:386-390)")
        self.ctx = ctx
        self.hmg = HipMG(ctx, levels, transfers, params["fieldsplit_0"], restriction=restriction, coarse_inv=coarse_inv)
        if not self.hmg.full:
            raise NotImplementedError("fieldsplit_0 must use pc_mg_type full (solver.py:366)")
        L = levels[-1]
        self.B, self.mass_diag = build_pressure_coupling(L)
        self.saddle = hip.Saddle(self.hmg.mg, self.B, self.mass_diag, L.nu, L.gamma, remove_constant_nullspace=True)
        self.rtol, self.atol = float(params.get("ksp_rtol", 1e-5)), float(params.get("ksp_atol", 1e-50))
        self.max_it = int(params.get("ksp_max_it", 10000))
        self.restart = int(params.get("ksp_gmres_restart", 30))
        self.n_u, self.n_p = self.saddle.n_u, self.saddle.n_p

    def solve(self, f, g=None):
        b = np.concatenate([np.asarray(f, dtype=np.float64), np.zeros(self.n_p) if g is None else np.asarray(g)])
        db, dx = self.ctx.vec(b), self.ctx.vec(self.n_u + self.n_p)
        its, rn = self.saddle.solve(db, dx, self.rtol, self.atol, self.max_it, self.restart)
        x = dx.get()
        return x[:self.n_u], x[self.n_u:], its, rn
