"""alfi_amd -- MI355X-native implementation of alfi's multigrid hot path (patch smoother, level SpMV, FGMRES(k) smoother,
Schoeberl transfers, PCMG cycles) behind alfi's plug-in surface.  Module paths follow the reference so that option values
like ``"alfi_amd.Star"`` resolve the way ``"alfi.Star"`` does (alfi/__init__.py:1-6)."""
from .relaxation import OrderedRelaxation, Star, MacroStar, Options, PlexLike      # noqa: F401
from .problem import (NavierStokesProblem, TwoDimLidDrivenCavityProblem,            # noqa: F401
                      ThreeDimLidDrivenCavityProblem, ThreeDimBackwardsFacingStepProblem, build_hierarchy)


def __getattr__(name):
    # the GPU-facing classes load libalfi_hip.so on first use (and fail loudly if it is missing)
    if name in ("HipPatchPC", "HipMG", "PC", "mg_levels_solver", "fieldsplit_0_mg", "DGMassInv", "outer_solver",
                "HipOuterSolver"):
        from . import solver
        return getattr(solver, name)
    if name in ("PkP0SchoeberlTransfer", "SVSchoeberlTransfer", "AutoSchoeberlTransfer", "CoarseCellPatches",
                "CoarseCellMacroPatches", "NullTransfer", "Constant", "Function"):
        from . import transfer
        return getattr(transfer, name)
    raise AttributeError(name)
