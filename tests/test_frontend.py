"""Front end with the reference's plug-in surface: option dictionaries, patch constructors (CPU) and, on the GPU,
HipPatchPC / HipMG / PkP0SchoeberlTransfer driven the way alfi drives PatchPC, PCMG and its transfer objects."""
import numpy as np
import pytest

import alfi_amd
from alfi_amd.problem import TwoDimLidDrivenCavityProblem, ThreeDimLidDrivenCavityProblem, build_hierarchy
from alfi_amd.relaxation import Options, PlexLike, Star, OrderedRelaxation, patch_points_to_dofs


class _FakePC(object):
    def __init__(self, L, options=None, prefix=""):
        self.level_data, self.options, self.prefix = L, options or {}, prefix
        self._dm = PlexLike(L.V.mesh)

    def getDM(self):
        return self._dm

    def getOptionsPrefix(self):
        return self.prefix


@pytest.mark.parametrize("mk,k", [(lambda: TwoDimLidDrivenCavityProblem(2), 2),
                                  (lambda: ThreeDimLidDrivenCavityProblem(1), 2)])
def test_python_star_constructor_equals_builtin_star(mk, k):
    lv, _ = build_hierarchy(mk(), 1, k, Re=10)
    L = lv[-1]
    pc = _FakePC(L)
    patches, iterset = Star()(pc)
    assert np.array_equal(iterset, np.arange(len(patches)))
    ptr, dofs, kept = patch_points_to_dofs(L.V, pc.getDM(), patches)
    # built-in patches are ordered by the seed's node number; python ones by vertex number
    got = {tuple(dofs[ptr[i]:ptr[i + 1]]) for i in range(len(ptr) - 1)}
    ref = {tuple(L.patch_dofs[L.patch_ptr[i]:L.patch_ptr[i + 1]]) for i in range(len(L.patch_ptr) - 1)}
    assert got == ref


def test_sort_order_grammar_and_sweeps():
    assert OrderedRelaxation.parse_sort_order("0+:1-|1+") == [[(0, 1), (1, -1)], [(1, 1)]]
    assert OrderedRelaxation.parse_sort_order("None") is None
    lv, _ = build_hierarchy(TwoDimLidDrivenCavityProblem(2), 1, 2, Re=10)
    L = lv[-1]
    pc = _FakePC(L, {"pc_patch_construction_Star_sort_order": "0+:1-"})
    patches, iterset = Star()(pc)
    assert sorted(iterset) == list(range(len(patches)))
    dm = pc.getDM()
    xy = np.array([dm.point_coords(dm.vStart + v) for v in range(L.V.mesh.num_vertices)])[iterset]
    key = list(zip(xy[:, 0], -xy[:, 1]))
    assert key == sorted(key)                                   # ascending x, then descending y (ldc2d.py:39)
    pc2 = _FakePC(L, {"pc_patch_construction_Star_sort_order": "0+|0-"})
    _, it2 = Star()(pc2)
    assert len(it2) == 2 * len(patches)                         # one permutation per '|' sweep (relaxation.py:145-147)
    # the reference's key functions close over the loop variable (relaxation.py:98-108): both sweeps of "0+|0-" use the LAST
    # key, i.e. the descending-x ordering twice; the per-sweep option gives the two different orderings
    n = len(patches)
    x2 = np.array([dm.point_coords(dm.vStart + v) for v in range(L.V.mesh.num_vertices)])[:, 0]
    assert np.array_equal(it2[:n], it2[n:]) and (np.diff(x2[it2[:n]]) <= 0).all()
    pc3 = _FakePC(L, {"pc_patch_construction_Star_sort_order": "0+|0-",
                      "pc_patch_construction_Star_sort_order_per_sweep": True})
    _, it3 = Star()(pc3)
    assert (np.diff(x2[it3[:n]]) >= 0).all() and (np.diff(x2[it3[n:]]) <= 0).all()


def test_option_dictionary_has_the_reference_keys():
    o = alfi_amd.mg_levels_solver(3)
    assert o["ksp_type"] == "fgmres" and o["ksp_max_it"] == 10 and o["ksp_convergence_test"] == "skip"
    assert o["pc_python_type"] == "alfi_amd.HipPatchPC"
    assert o["patch_pc_patch_construct_type"] == "star" and o["patch_pc_patch_construct_dim"] == 0
    assert o["patch_pc_patch_dense_inverse"] is True and o["patch_pc_patch_sub_mat_type"] == "seqdense"
    assert alfi_amd.mg_levels_solver(2)["ksp_max_it"] == 6
    m = alfi_amd.mg_levels_solver(2, patch_composition="multiplicative", relaxation_direction="0+:1-")
    assert m["patch_pc_patch_construct_python_type"] == "alfi_amd.Star"
    assert m["patch_pc_patch_construction_Star_sort_order"] == "0+:1-"
    with pytest.raises(NotImplementedError):
        alfi_amd.mg_levels_solver(2, patch_composition="multiplicative")
    f = alfi_amd.fieldsplit_0_mg(o)
    assert f["pc_mg_type"] == "full" and f["ksp_type"] == "richardson" and f["ksp_max_it"] == 1


@pytest.mark.gpu
def test_hip_patch_pc_and_mg_from_options():
    from alfi_amd import hip
    from oracle import alfi_oracle as O
    lv, tr = build_hierarchy(ThreeDimLidDrivenCavityProblem(2), 1, 1, Re=100.0)
    ctx = hip.Context(0)
    params = alfi_amd.fieldsplit_0_mg(alfi_amd.mg_levels_solver(3, smoothing=3))
    mg = alfi_amd.HipMG(ctx, lv, tr, params)
    L = lv[-1]
    rng = np.random.default_rng(0)
    # PCPython protocol: apply(pc, x, y) with host arrays
    x, y = rng.standard_normal(L.n), np.zeros(L.n)
    mg.pc_objs[-1].apply(mg.pcs[-1], x, y)
    omg = O.build_oracle_mg(lv, tr, 3)
    ref = omg.levels[-1]["smoother"].apply(x)
    assert np.abs(y - ref).max() < 1e-7 * np.abs(ref).max()
    with pytest.raises(NotImplementedError):
        mg.pc_objs[-1].applyTranspose(mg.pcs[-1], x, y)
    # one PCMG(full) application
    b = rng.standard_normal(L.n)
    b[L.bc_dofs] = 0
    out = np.zeros(L.n)
    mg.apply(b, out)
    ref = omg.fcycle(b)
    assert np.abs(out - ref).max() < 1e-5 * np.abs(ref).max()
    # python-constructed Star patches give the same preconditioner as the built-in star
    o2 = alfi_amd.mg_levels_solver(3, smoothing=3)
    o2["patch_pc_patch_construct_type"] = "python"
    o2["patch_pc_patch_construct_python_type"] = "alfi_amd.Star"
    pc = alfi_amd.PC(ctx, L, options=o2)
    obj = alfi_amd.HipPatchPC()
    obj.initialize(pc)
    y2 = np.zeros(L.n)
    obj.apply(pc, x, y2)
    assert np.abs(y2 - y).max() < 1e-10 * np.abs(y).max()
    # unsupported modes are refused loudly
    o3 = alfi_amd.mg_levels_solver(3, smoothing=3)
    o3["patch_pc_patch_sub_mat_type"] = "baij"
    with pytest.raises(NotImplementedError):
        alfi_amd.HipPatchPC().initialize(alfi_amd.PC(ctx, L, options=o3))


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [2, 3])
def test_multiplicative_sweeps_through_the_option_dictionary(dim):
    """--patch-composition multiplicative (solver.py:306, 322-324, 332-335): python-constructed Star patches ordered by
    the problem's relaxation_direction, symmetrised sweep; PatchPC.apply and the whole PCMG against the oracle."""
    import alfi_amd
    from alfi_amd import hip
    from oracle import alfi_oracle as O
    prob = TwoDimLidDrivenCavityProblem(4) if dim == 2 else ThreeDimLidDrivenCavityProblem(2)
    lv, tr = build_hierarchy(prob, 1, 2, Re=100.0)
    ctx = hip.Context(0)
    L = lv[-1]
    opts = alfi_amd.mg_levels_solver(dim, patch_composition="multiplicative", smoothing=3,
                                     relaxation_direction=prob.relaxation_direction())
    assert opts["patch_pc_patch_local_type"] == "multiplicative" and opts["patch_pc_patch_symmetrise_sweep"]
    pc = alfi_amd.PC(ctx, L, options=opts)
    obj = alfi_amd.HipPatchPC()
    obj.initialize(pc)
    assert obj.wavefronts > 1 and len(obj.iterset) == len(obj.patch_ptr) - 1
    x = np.random.default_rng(3).standard_normal(L.n)
    y = np.zeros(L.n)
    obj.apply(pc, x, y)
    sm = O.PatchSmoother(L.A.to_scipy().tocsr(), obj.patch_ptr, obj.patch_dofs, L.bc_dofs, "multiplicative",
                         obj.iterset, True)
    ref = sm.apply(x)
    assert np.abs(y - ref).max() < 1e-7 * np.abs(ref).max()
    # the sweep order matters: the reversed order gives a different (but equally valid) preconditioner
    sm_rev = O.PatchSmoother(L.A.to_scipy().tocsr(), obj.patch_ptr, obj.patch_dofs, L.bc_dofs, "multiplicative",
                             obj.iterset[::-1], False)
    assert np.abs(sm_rev.apply(x) - ref).max() > 1e-3 * np.abs(ref).max()
    # whole multigrid with the multiplicative smoother
    mg = alfi_amd.HipMG(ctx, lv, tr, alfi_amd.fieldsplit_0_mg(opts))
    b = np.random.default_rng(4).standard_normal(L.n)
    b[L.bc_dofs] = 0.0
    out = np.zeros(L.n)
    mg.apply(b, out)
    itersets = [None] + [o.iterset for o in mg.pc_objs[1:]]
    olev_patches = [None] + [(o.patch_ptr, o.patch_dofs) for o in mg.pc_objs[1:]]
    for Lv, pp in zip(lv, olev_patches):
        if pp is not None:
            Lv.patch_ptr, Lv.patch_dofs = pp
    omg = O.build_oracle_mg(lv, tr, 3, local_type="multiplicative", itersets=itersets, symmetrise=True)
    ref = omg.fcycle(b)
    assert np.abs(out - ref).max() < 1e-5 * np.abs(ref).max()


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [2, 3])
def test_multiplicative_sweeps_over_macro_star_patches(dim):
    """--patch macro --patch-composition multiplicative (alfi/solver.py:322-324 with 339-342): MacroStar patches (74 dofs =
    37 nodes in 2-D, up to 1533 dofs = 511 nodes in 3-D, i.e. beyond the 64 nodes a wave sweeps) ordered by the problem's
    relaxation_direction, symmetrised sweep with the workgroup-per-patch kernel; PatchPC.apply and the whole PCMG against
    the oracle's sequential sweep."""
    import alfi_amd
    from alfi_amd import hip
    from oracle import alfi_oracle as O
    prob = TwoDimLidDrivenCavityProblem(4) if dim == 2 else ThreeDimLidDrivenCavityProblem(2)
    lv, tr = build_hierarchy(prob, 1, 2, Re=100.0)
    ctx = hip.Context(0)
    L = lv[-1]
    opts = alfi_amd.mg_levels_solver(dim, patch="macro", patch_composition="multiplicative", smoothing=2,
                                     relaxation_direction=prob.relaxation_direction())
    assert opts["patch_pc_patch_construction_MacroStar_sort_order"] == "0+:1-"
    pc = alfi_amd.PC(ctx, L, options=opts)
    obj = alfi_amd.HipPatchPC()
    obj.initialize(pc)
    sizes = np.diff(obj.patch_ptr)
    assert sizes.max() == (74 if dim == 2 else 1533) and obj.wavefronts >= 1
    x = np.random.default_rng(3).standard_normal(L.n)
    y = np.zeros(L.n)
    obj.apply(pc, x, y)
    sm = O.PatchSmoother(L.A.to_scipy().tocsr(), obj.patch_ptr, obj.patch_dofs, L.bc_dofs, "multiplicative",
                         obj.iterset, True)
    ref = sm.apply(x)
    assert np.abs(y - ref).max() < 1e-7 * np.abs(ref).max()
    add = O.PatchSmoother(L.A.to_scipy().tocsr(), obj.patch_ptr, obj.patch_dofs, L.bc_dofs).apply(x)
    assert np.abs(add - ref).max() > 1e-3 * np.abs(ref).max()        # it is not the additive operator
    mg = alfi_amd.HipMG(ctx, lv, tr, alfi_amd.fieldsplit_0_mg(opts))
    b = np.random.default_rng(4).standard_normal(L.n)
    b[L.bc_dofs] = 0.0
    out = np.zeros(L.n)
    mg.apply(b, out)
    itersets = [None] + [o.iterset for o in mg.pc_objs[1:]]
    for Lv, o in zip(lv[1:], mg.pc_objs[1:]):
        Lv.patch_ptr, Lv.patch_dofs = o.patch_ptr, o.patch_dofs
    ref = O.build_oracle_mg(lv, tr, 2, local_type="multiplicative", itersets=itersets, symmetrise=True).fcycle(b)
    assert np.abs(out - ref).max() < 1e-5 * np.abs(ref).max()
    mg.mg.close()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [2, 3])
def test_partition_of_unity_and_stars_of_edges(dim):
    """Two PCPATCH options the reference never changes from their defaults (solver.py:321, 338) but a drop-in accepts:
    ``patch_pc_patch_partition_of_unity: True`` (additive sums weighted by 1 / multiplicity) and the built-in star with
    ``patch_pc_patch_construct_dim: 1`` (one patch per edge: the edge and the faces around it)."""
    import alfi_amd
    from alfi_amd import hip
    from oracle import alfi_oracle as O
    prob = TwoDimLidDrivenCavityProblem(4) if dim == 2 else ThreeDimLidDrivenCavityProblem(2)
    lv, _ = build_hierarchy(prob, 1, 2, Re=100.0)
    L = lv[-1]
    ctx = hip.Context(0)
    x = np.random.default_rng(9).standard_normal(L.n)
    A = L.A.to_scipy().tocsr()
    opts = alfi_amd.mg_levels_solver(dim, smoothing=2)
    opts["patch_pc_patch_partition_of_unity"] = True
    pc = alfi_amd.PC(ctx, L, options=opts)
    obj = alfi_amd.HipPatchPC()
    obj.initialize(pc)
    y = np.zeros(L.n)
    obj.apply(pc, x, y)
    ref = O.PatchSmoother(A, obj.patch_ptr, obj.patch_dofs, L.bc_dofs, partition_of_unity=True).apply(x)
    assert np.abs(y - ref).max() < 1e-7 * np.abs(ref).max()
    plain = O.PatchSmoother(A, obj.patch_ptr, obj.patch_dofs, L.bc_dofs).apply(x)
    assert np.abs(plain - ref).max() > 1e-2 * np.abs(ref).max()
    obj.level.close()
    opts = alfi_amd.mg_levels_solver(dim, smoothing=2)
    opts["patch_pc_patch_construct_dim"] = 1
    pc = alfi_amd.PC(ctx, L, options=opts)
    obj = alfi_amd.HipPatchPC()
    obj.initialize(pc)
    sizes = np.diff(obj.patch_ptr)
    m = L.V.mesh
    nfree_edges = int(np.count_nonzero(~L.V.bc_node_mask[np.asarray(L.V.edge_nodes).ravel()]))
    if dim == 2:
        assert len(sizes) == nfree_edges and (sizes == 2).all()           # an edge star holds the edge's own P2 node
    else:
        assert sizes.max() == 3 * 7 and len(sizes) >= nfree_edges         # edge node + the 6 face bubbles around an interior edge
    y = np.zeros(L.n)
    obj.apply(pc, x, y)
    ref = O.PatchSmoother(A, obj.patch_ptr, obj.patch_dofs, L.bc_dofs).apply(x)
    assert np.abs(y - ref).max() < 1e-7 * np.abs(ref).max()
    obj.level.close()
    ctx.close()


@pytest.mark.gpu
def test_schoeberl_transfer_object_protocol():
    from alfi_amd import hip, Constant, Function, PkP0SchoeberlTransfer
    from oracle import alfi_oracle as O
    lv, tr = build_hierarchy(TwoDimLidDrivenCavityProblem(4), 1, 2, Re=100.0)
    nu, gamma = Constant(lv[-1].nu), Constant(lv[-1].gamma)
    vt = PkP0SchoeberlTransfer((nu, gamma), 2, "uniform")
    Vc, Vf = lv[0].V, lv[1].V
    rng = np.random.default_rng(1)
    uc = rng.standard_normal(Vc.num_dofs)
    uc[lv[0].bc_dofs] = 0
    coarse, fine = Function(Vc, uc), Function(Vf)
    vt.prolong(coarse, fine)
    ot = O.oracle_transfer(tr[-1], lv[-1], True).st
    ref = ot.prolong(uc)
    ref[lv[1].bc_dofs] = 0
    assert np.abs(fine.dat.data.ravel() - ref).max() < 1e-8 * np.abs(ref).max()
    r = rng.standard_normal(Vf.num_dofs)
    fr, cr = Function(Vf, r), Function(Vc)
    vt.restrict(fr, cr)
    ref = ot.restrict(r)
    ref[lv[0].bc_dofs] = 0
    assert np.abs(cr.dat.data.ravel() - ref).max() < 1e-8 * np.abs(ref).max()
    assert np.array_equal(fr.dat.data.ravel(), r)
    # inject (solver.py:595): nodal values at the coarse nodes; inject(plain prolongation) is the identity
    fi, ci = Function(Vf, tr[-1].PT_plain.to_scipy().T @ uc), Function(Vc)
    vt.inject(fi, ci)
    assert np.abs(ci.dat.data.ravel() - uc).max() < 1e-13
    assert np.array_equal(ci.dat.data, fi.dat.data[tr[-1].inject_map])
    # changing gamma triggers a rebuild of the interior solves (transfer.py:173-184)
    gamma.assign(10.0)
    vt.prolong(coarse, fine)
    tr2 = build_hierarchy(TwoDimLidDrivenCavityProblem(4), 1, 2, Re=100.0, gamma=10.0)
    ot2 = O.oracle_transfer(tr2[1][-1], tr2[0][-1], True).st
    ref2 = ot2.prolong(uc)
    ref2[lv[1].bc_dofs] = 0
    assert np.abs(fine.dat.data.ravel() - ref2).max() < 1e-8 * np.abs(ref2).max()
    vt.break_ref_cycles()


@pytest.mark.gpu
def test_sv_schoeberl_transfer_object_protocol():
    """SVSchoeberlTransfer((nu, gamma), tdim, "bary") (transfer.py:293-309) with the reference's prolong / restrict / inject
    protocol on a barycentric hierarchy; PkP0SchoeberlTransfer refuses that hierarchy and vice versa."""
    from alfi_amd import Constant, Function, PkP0SchoeberlTransfer, SVSchoeberlTransfer, CoarseCellMacroPatches
    from alfi_amd.sv import build_sv_hierarchy
    from oracle import alfi_oracle as O
    lv, tr = build_sv_hierarchy(TwoDimLidDrivenCavityProblem(2), 1, 2, Re=100.0)
    nu, gamma = Constant(lv[-1].nu), Constant(lv[-1].gamma)
    with pytest.raises(NotImplementedError):
        PkP0SchoeberlTransfer((nu, gamma), 2, "bary")
    with pytest.raises(NotImplementedError):
        SVSchoeberlTransfer((nu, gamma), 2, "uniform")
    vt = SVSchoeberlTransfer((nu, gamma), 2, "bary")
    Vc, Vf = lv[0].V, lv[1].V
    rng = np.random.default_rng(1)
    uc = rng.standard_normal(Vc.num_dofs)
    uc[lv[0].bc_dofs] = 0
    coarse, fine = Function(Vc, uc), Function(Vf)
    vt.prolong(coarse, fine)
    ot = O.oracle_transfer(tr[-1], lv[-1], True).st
    ref = ot.prolong(uc)
    ref[lv[1].bc_dofs] = 0
    assert np.abs(fine.dat.data.ravel() - ref).max() < 1e-8 * np.abs(ref).max()
    r = rng.standard_normal(Vf.num_dofs)
    fr, cr = Function(Vf, r), Function(Vc)
    vt.restrict(fr, cr)
    ref = ot.restrict(r)
    ref[lv[0].bc_dofs] = 0
    assert np.abs(cr.dat.data.ravel() - ref).max() < 1e-8 * np.abs(ref).max()
    # inject on the non-nested hierarchy: point evaluation; inject(prolong(P2 field)) is the identity
    X = Vc.node_coords
    q = np.stack([X[:, 0] ** 2 - X[:, 1], X[:, 0] * X[:, 1]], axis=1)
    fi, ci = Function(Vf, tr[-1].P.to_scipy() @ q.ravel()), Function(Vc)
    vt.inject(fi, ci)
    assert np.abs(ci.dat.data - q).max() < 1e-12
    # ... applied ON THE DEVICE (alfi_transfer_set_injection_matrix): equal to the host product with sv.bary_injection
    rnd = Function(Vf, np.random.default_rng(3).standard_normal(Vf.num_dofs))
    vt.inject(rnd, ci)
    ref = tr[-1].inject_matrix @ rnd.dat.data
    assert np.abs(ci.dat.data - ref).max() < 1e-14 * np.abs(ref).max()

    class _PC(object):
        level_data = lv[1]
    blocks, it = CoarseCellMacroPatches()(_PC())
    assert np.array_equal(np.asarray(blocks), tr[-1].blk_dofs[:, ::2] // 2)
    vt.break_ref_cycles()


@pytest.mark.gpu
def test_macro_star_patches_through_the_option_dictionary():
    """``patch="macro"`` (solver.py:339-343): python-constructed ``MacroStar`` patches (relaxation.py:163-177) -- one per
    MacroVertices-labelled vertex, the union of the stars of every vertex of its macro neighbourhood; 74 dofs for an
    interior macro vertex of a once-refined ldc2d mesh -- gathered, inverted and applied on the GPU, against the
    oracle's dense patch solves on the same dof sets."""
    import alfi_amd
    from alfi_amd import hip
    from oracle import alfi_oracle as O
    lv, tr = build_hierarchy(TwoDimLidDrivenCavityProblem(4), 1, 2, Re=100.0)
    L = lv[-1]
    ctx = hip.Context(0)
    opts = alfi_amd.mg_levels_solver(2, patch="macro", smoothing=3)
    assert opts["patch_pc_patch_construct_python_type"] == "alfi_amd.MacroStar"
    pc = alfi_amd.PC(ctx, L, options=opts)
    obj = alfi_amd.HipPatchPC()
    obj.initialize(pc)
    sizes = np.diff(obj.patch_ptr)
    assert len(sizes) == 25 and sizes.max() == 74                 # 5 x 5 macro vertices
    x = np.random.default_rng(5).standard_normal(L.n)
    y = np.zeros(L.n)
    obj.apply(pc, x, y)
    ref = O.PatchSmoother(L.A.to_scipy().tocsr(), obj.patch_ptr, obj.patch_dofs, L.bc_dofs).apply(x)
    assert np.abs(y - ref).max() < 1e-8 * np.abs(ref).max()
    obj.level.close()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k,nmax", [(1, 1053), (2, 1533)])
def test_macro_star_patches_3d_use_the_large_patch_path(k, nmax):
    """3-D macro stars (solver.py:339-343 / relaxation.py:163-177) hold 60 .. 1533 dofs on a once-refined ldc3d mesh -- the
    size class of the reference's Scott-Vogelius macro patches (405 / 1275): blocked matrix-core
    inversion + workgroup-per-patch apply behind the same PCPython class; apply and a whole multigrid cycle with that
    smoother against the oracle."""
    import alfi_amd
    from alfi_amd import hip
    from oracle import alfi_oracle as O
    lv, tr = build_hierarchy(ThreeDimLidDrivenCavityProblem(2), 1, k, Re=100.0)
    L = lv[-1]
    ctx = hip.Context(0)
    opts = alfi_amd.mg_levels_solver(3, patch="macro", smoothing=3)
    pc = alfi_amd.PC(ctx, L, options=opts)
    obj = alfi_amd.HipPatchPC()
    obj.initialize(pc)
    sizes = np.diff(obj.patch_ptr)
    assert len(sizes) == 27 and sizes.max() == nmax
    x = np.random.default_rng(7).standard_normal(L.n)
    y = np.zeros(L.n)
    obj.apply(pc, x, y)
    sm = O.PatchSmoother(L.A.to_scipy().tocsr(), obj.patch_ptr, obj.patch_dofs, L.bc_dofs)
    ref = sm.apply(x)
    # the blocked inversion is followed by a Newton-Schulz polish, so the macro patches meet the tolerance of the stars
    assert np.abs(y - ref).max() < 1e-7 * np.abs(ref).max()
    obj.level.close()
    mg = alfi_amd.HipMG(ctx, lv, tr, alfi_amd.fieldsplit_0_mg(opts))
    b = np.random.default_rng(8).standard_normal(L.n)
    b[L.bc_dofs] = 0.0
    out = np.zeros(L.n)
    mg.apply(b, out)
    lv[-1].patch_ptr, lv[-1].patch_dofs = mg.pc_objs[-1].patch_ptr, mg.pc_objs[-1].patch_dofs
    ref = O.build_oracle_mg(lv, tr, 3).fcycle(b)
    assert np.abs(out - ref).max() < 1e-5 * np.abs(ref).max()
    mg.mg.close()
    ctx.close()


def test_macro_star_on_an_alfeld_split_mesh():
    """The reference's macro-star patches live on barycentrically refined meshes (bary.py:16-27; MacroVertices = the
    vertices of the mesh that is split).  For a fully interior macro vertex of a Kuhn box and [P2]^3:
      * 24 macro tets -> 96 cells; the interior of the macro patch holds 25 vertices + 110 edges = 135 nodes = 405 dofs --
        the figure of SURVEY.md section 8 (and 425 nodes = 1275 dofs for P3);
      * the LITERAL callback (relaxation.py:168-177) tests the MacroVertices label on every point of the closures, not only
        on vertices, so the 36 outer edges of the macro star bring their own stars -- and their dofs -- along: 513 dofs.
        alfi_amd.MacroStar follows the literal code; both sizes are within the library's 4096-dof limit."""
    from alfi_amd.mesh import box_mesh, bary_refine, bary_mesh_hierarchy, bfs3d_mesh
    from alfi_amd.elements import NodalElement
    from alfi_amd.fespace import VectorFunctionSpace
    from alfi_amd.relaxation import MacroStar
    m = box_mesh(4, 4, 4, 2.0, 2.0, 2.0)
    b = bary_refine(m)
    assert b.num_cells == 4 * m.num_cells and b.num_vertices == m.num_vertices + m.num_cells
    assert np.isclose(b.cell_geometry()[1].sum(), 8.0) and b.macro_vertex_mask.sum() == m.num_vertices
    assert np.array_equal(b.parent_cell, np.repeat(np.arange(m.num_cells), 4))
    V = VectorFunctionSpace(b, NodalElement(3, 2, False))
    dm = PlexLike(b, labels={"MacroVertices": {int(b.num_cells + v): 1 for v in np.flatnonzero(b.macro_vertex_mask)}})
    vid = int(np.flatnonzero((np.abs(m.coords - 1.0) < 1e-12).all(axis=1))[0])        # the centre vertex
    ms = MacroStar()
    assert ms.callback(dm, dm.vStart + m.num_vertices) is None                          # a barycentre is no seed
    pts = ms.callback(dm, dm.vStart + vid)
    ptr, dofs, _ = patch_points_to_dofs(V, dm, [np.asarray(pts)])
    assert np.diff(ptr)[0] == 513
    s = list(ms.star(dm, dm.vStart + vid))
    clos = sum((list(ms.closure(dm, e)) for e in s), [])
    vs = [p for p in set(clos) if dm.vStart <= p < dm.eStart and dm.getLabelValue("MacroVertices", p) != 1]
    assert len(vs) == 24                                                                # the 24 barycentres
    ptr2, _, _ = patch_points_to_dofs(V, dm, [np.asarray(s + sum((list(ms.star(dm, v)) for v in vs), []))])
    assert np.diff(ptr2)[0] == 405
    # hierarchy and the step channel of config 5
    mh = bary_mesh_hierarchy(box_mesh(1, 1, 1, 1.0, 1.0, 1.0), 1)
    assert [x.num_cells for x in mh] == [24, 192] and mh[1].macro_mesh.num_cells == 48
    step = bfs3d_mesh(2)
    assert np.isclose(step.cell_geometry()[1].sum(), 10 * 2 * 1 - 1.0)
