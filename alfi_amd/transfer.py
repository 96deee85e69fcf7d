"""Grid transfers with the reference's object protocol (alfi/transfer.py): ``PkP0SchoeberlTransfer((nu, gamma), tdim,
hierarchy)`` with ``prolong(coarse, fine)`` / ``restrict(fine, coarse)`` writing into their second argument, caches per
fine space keyed by ``V.dim()`` and a rebuild of the interior solves when ``float(nu)`` or ``float(gamma)`` changed
(transfer.py:173-184, 238-244); ``NullTransfer`` for the pressure (transfer.py:359-366).

The arithmetic runs in libalfi_hip.so (``alfi_prolong`` / ``alfi_restrict``).  What the reference re-does on every call
-- a full finite-element assembly of ``bform`` (transfer.py:249, 272) -- is a block-CSR SpMV with precomputed interior
rows of the grad-div matrix here.
"""
import numpy as np

from . import hip
from .problem import BSR, build_transfer_data
from .fespace import coarse_cell_blocks


class Constant(object):
    """Mutable scalar, like firedrake.Constant: ``float(c)``, ``c.assign(v)``."""

    def __init__(self, value):
        self.value = float(value)

    def assign(self, value):
        self.value = float(value)
        return self

    def __float__(self):
        return self.value


class Function(object):
    """A coefficient vector on a VectorFunctionSpace: ``f.dat.data`` is the host array (shape (nodes, dim))."""

    class _Dat(object):
        def __init__(self, data):
            self.data = data

        @property
        def data_ro(self):
            return self.data

    def __init__(self, V, data=None):
        self.V = V
        self.dat = Function._Dat(np.zeros((V.num_nodes, V.dim)) if data is None
                                 else np.array(data, dtype=np.float64).reshape(V.num_nodes, V.dim))

    def function_space(self):
        return self.V

    @property
    def ufl_shape(self):
        return (self.V.dim,)


class CoarseCellPatches(object):
    """transfer.py:13-46: one patch per coarse cell = fine points in the closure of its children that are not on the
    coarse skeleton.  Here directly as interior node blocks (``fespace.coarse_cell_blocks``)."""

    def __call__(self, pc):
        V = pc.level_data.V
        blocks = coarse_cell_blocks(V)
        return [b for b in blocks], np.arange(blocks.shape[0], dtype=np.int64)


class CoarseCellMacroPatches(object):
    """transfer.py:49-88: on a barycentric hierarchy one patch per coarse MACRO cell = the fine points strictly inside it
    (``sv.macro_cell_blocks``)."""

    def __call__(self, pc):
        from .sv import macro_cell_blocks
        blocks = macro_cell_blocks(pc.level_data.V)
        return [b for b in blocks], np.arange(blocks.shape[0], dtype=np.int64)


class AutoSchoeberlTransfer(object):
    hierarchies = ("uniform",)

    def __init__(self, parameters, tdim, hierarchy, ctx=None):
        if hierarchy not in self.hierarchies:
            raise NotImplementedError("%s on a %r hierarchy (PkP0SchoeberlTransfer: uniform; SVSchoeberlTransfer: bary)"
                                      % (type(self).__name__, hierarchy))
        self.parameters = parameters
        self.tdim = tdim
        self.ctx = ctx
        self.solver = {}            # key -> (device transfer, shell levels)
        self.prev_parameters = {}
        self.force_rebuild_d = {}

    def break_ref_cycles(self):
        for dev, levels in self.solver.values():
            dev.close()
            for l in levels:
                l.close()
        self.solver = {}

    def force_rebuild(self):
        self.force_rebuild_d = {k: True for k in self.prev_parameters}

    def rebuild(self, key):
        if self.force_rebuild_d.get(key, False):
            self.force_rebuild_d[key] = False
            return True
        prev = self.prev_parameters.get(key, [])
        return any(float(p) != q for (q, p) in zip(prev, self.parameters))

    def prolong(self, coarse, fine):
        self.restrict_or_prolong(coarse, fine, "prolong")

    def restrict(self, fine, coarse):
        self.restrict_or_prolong(fine, coarse, "restrict")

    def _setup(self, Vc, Vf):
        if self.ctx is None:
            self.ctx = hip.Context(0)
        nu, gamma = (float(p) for p in self.parameters)
        T = self._transfer_data(Vc, Vf, nu, gamma)

        def shell(V):       # a level without operator: only sizes and Dirichlet dofs are needed by the transfer
            empty = BSR(V.num_nodes, V.num_nodes, V.dim, np.zeros(V.num_nodes + 1, dtype=np.int32),
                        np.zeros(0, dtype=np.int32), np.zeros((0, V.dim, V.dim)))
            return hip.Level(self.ctx, empty, V.bc_dofs)
        lc, lf = shell(Vc), shell(Vf)
        dev = hip.Transfer(self.ctx, lc, lf, T)
        dev.update(nu, gamma)
        return dev, (lc, lf)

    def _transfer_data(self, Vc, Vf, nu, gamma):
        return build_transfer_data(Vc, Vf, nu, gamma)

    def restrict_or_prolong(self, source, target, mode):
        coarse, fine = (source, target) if mode == "prolong" else (target, source)
        Vc, Vf = coarse.function_space(), fine.function_space()
        key = Vf.num_dofs
        if key not in self.solver:
            self.solver[key] = self._setup(Vc, Vf)
            self.prev_parameters[key] = [float(p) for p in self.parameters]
        elif self.rebuild(key):
            self.solver[key][0].update(*(float(p) for p in self.parameters))
            self.prev_parameters[key] = [float(p) for p in self.parameters]
        dev = self.solver[key][0]
        ctx = self.ctx
        if mode == "prolong":
            dxc, dxf = ctx.vec(coarse.dat.data.ravel()), ctx.vec(Vf.num_dofs)
            dev.prolong(dxc, dxf)
            fine.dat.data[:] = dxf.get().reshape(fine.dat.data.shape)
        else:
            drf, drc = ctx.vec(fine.dat.data.ravel()), ctx.vec(Vc.num_dofs)
            dev.restrict(drf, drc, robust=True)
            coarse.dat.data[:] = drc.get().reshape(coarse.dat.data.shape)


    def inject(self, fine, coarse):
        """firedrake.inject for the nodal velocity spaces (the third entry of the transfer tuple, solver.py:595)."""
        Vc, Vf = coarse.function_space(), fine.function_space()
        key = Vf.num_dofs
        if key not in self.solver:
            self.solver[key] = self._setup(Vc, Vf)
            self.prev_parameters[key] = [float(p) for p in self.parameters]
        dev, ctx = self.solver[key][0], self.ctx
        # (non-nested bary hierarchy: the device applies the point-evaluation matrix, alfi_transfer_set_injection_matrix)
        dxf, dxc = ctx.vec(fine.dat.data.ravel()), ctx.vec(Vc.num_dofs)
        dev.inject(dxf, dxc)
        coarse.dat.data[:] = dxc.get().reshape(coarse.dat.data.shape)


class PkP0SchoeberlTransfer(AutoSchoeberlTransfer):
    """transfer.py:312-356: forms nu (2 sym grad u, grad v) + gamma (cell_avg div u, div v); the standard transfer is
    the bubble transfer for 3-D P1+FB and nodal interpolation otherwise (decided in fespace.vector_prolongation)."""


class SVSchoeberlTransfer(AutoSchoeberlTransfer):
    """transfer.py:293-309: Scott-Vogelius forms nu (2 sym grad u, grad v) + gamma (div u, div v) on a barycentric hierarchy;
    interior blocks = coarse macro cells (CoarseCellMacroPatches), standard transfer = firedrake's non-nested prolong
    (``sv.bary_prolongation``)."""
    hierarchies = ("bary",)

    def _transfer_data(self, Vc, Vf, nu, gamma):
        from . import _hostlib
        from .sv import build_sv_transfer_data
        return build_sv_transfer_data(Vc, Vf, nu, gamma, _hostlib.node_graph(Vf.cell_nodes, Vf.num_nodes))


class NullTransfer(object):
    """transfer.py:359-366: pressure is never needed on coarse levels."""

    def transfer(self, src, dest):
        dest.dat.data[...] = np.nan

    inject = transfer
    prolong = transfer
    restrict = transfer
