"""Rank-local generation (alfi_amd.lazy) against cutting the rank's share out of the global hierarchy: same partition, same
ghosts, same local operators, patches and transfer pieces for every rank.  CPU only (host generator + partitioner)."""
import numpy as np
import pytest

from alfi_amd import dist as D
from alfi_amd.problem import TwoDimLidDrivenCavityProblem, ThreeDimLidDrivenCavityProblem, build_hierarchy

CASES = {
    "2d-P2": (lambda: TwoDimLidDrivenCavityProblem(4), 2, 2, 100.0, 3, 200),
    "3d-P2FB": (lambda: ThreeDimLidDrivenCavityProblem(2), 2, 2, 100.0, 2, 2000),
    "3d-P1FB-bubble": (lambda: ThreeDimLidDrivenCavityProblem(2), 2, 1, 100.0, 3, 1000),
}


def _same_bsr(a, b, tol=1e-13):
    assert a.nbrows == b.nbrows and a.nbcols == b.nbcols and a.bs == b.bs
    assert np.array_equal(a.rowptr, b.rowptr) and np.array_equal(a.colidx, b.colidx)
    scale = max(np.abs(b.vals).max(), 1e-300) if b.vals.size else 1.0
    assert np.abs(a.vals - b.vals).max() <= tol * scale if b.vals.size else True


@pytest.mark.parametrize("case", sorted(CASES))
def test_rank_local_generation_equals_cut_from_global(case):
    mk, nref, k, Re, world, min_dofs = CASES[case]
    glv, gtr = build_hierarchy(mk(), nref, k, Re=Re)
    llv, ltr = build_hierarchy(mk(), nref, k, Re=Re, lazy=True)
    assert all(getattr(L.A, "is_lazy", False) for L in llv) and all(T.is_lazy for T in ltr)
    gs = D.choose_splits(glv, world, min_dofs)
    ls = D.choose_splits(llv, world, min_dofs)
    for a, b in zip(gs, ls):
        assert np.array_equal(a, b)
    assert any(np.count_nonzero(np.diff(s)) > 1 for s in gs), "the case must exercise a distributed level"
    for rank in range(world):
        gp = D.build_parts(glv, gtr, gs, rank)
        lp = D.build_parts(llv, ltr, ls, rank)
        for a, b in zip(gp, lp):
            assert np.array_equal(a.ghosts, b.ghosts) and np.array_equal(a.own_nodes, b.own_nodes)
            for x, y in zip(a.send_nodes, b.send_nodes):
                assert np.array_equal(x, y)
        gl, gt, gmin = D.localize(glv, gtr, gp)
        ll, lt, lmin = D.localize(llv, ltr, lp)
        assert gmin == lmin and len(gl) == len(ll) and len(gt) == len(lt)
        for a, b in zip(gl, ll):
            _same_bsr(b.A, a.A)
            assert np.array_equal(a.bc_dofs, b.bc_dofs)
            assert np.array_equal(a.patch_ptr, b.patch_ptr) and np.array_equal(a.patch_dofs, b.patch_dofs)
            assert a.npatch_int == b.npatch_int
        for a, b in zip(gt, lt):
            assert np.array_equal(a.blocks, b.blocks) and np.array_equal(a.blk_dofs, b.blk_dofs)
            for name in ("K_II", "D_II"):
                x, y = getattr(a, name), getattr(b, name)
                assert x.shape == y.shape and np.abs(x - y).max() <= 1e-13 * np.abs(x).max()
            for name in ("D_I", "D_IT", "P", "PT", "PT_plain"):
                _same_bsr(getattr(b, name), getattr(a, name))


def test_lazy_operator_materialises_to_the_global_one():
    glv, _ = build_hierarchy(ThreeDimLidDrivenCavityProblem(2), 1, 2, Re=100.0)
    llv, _ = build_hierarchy(ThreeDimLidDrivenCavityProblem(2), 1, 2, Re=100.0, lazy=True)
    for g, l in zip(glv, llv):
        _same_bsr(l.A.materialise(), g.A)
        rows = np.arange(g.A.nbrows)[::-3]                      # any order, any subset
        _same_bsr(l.A.select_rows(rows), g.A.select_rows(rows))
