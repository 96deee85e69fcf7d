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


def _shared_worker(rank, world, port, q):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from alfi_amd.problem import ThreeDimLidDrivenCavityProblem, build_hierarchy
        import glob
        from alfi_amd.shared import build_shared, _base
        calls, names = [], []

        def build():
            calls.append(1)
            return build_hierarchy(ThreeDimLidDrivenCavityProblem(2), 1, 2, Re=100.0, lazy=True)

        def bcast(x):
            box = [x]
            dist.broadcast_object_list(box, src=0)
            names.append(box[0][1])
            return box[0]
        lv, tr = build_shared(build, rank, bcast, dist.barrier, "test_%d" % port)
        L = lv[-1]
        own = L.A.select_rows(np.arange(rank, L.A.nbrows, world))       # rank-local values from the shared integer side
        # copy-on-write: a write by one rank is private to it
        L.patch_dofs[0] = -7 - rank
        dist.barrier()
        left = os.path.exists(names[0]) or bool(glob.glob(os.path.join(_base(), "alfi_gen_test_%d_*" % port)))
        q.put((rank, len(calls), int(L.n), float(np.abs(own.vals).sum()), int(L.patch_dofs[0]), left))
    finally:
        dist.destroy_process_group()


def test_hierarchy_generated_once_and_shared_between_ranks():
    """alfi_amd.shared (gloo, world 3): rank 0 generates, the others map the file copy-on-write; every rank can assemble its
    own rows from the shared integer side, writes stay private, and nothing is left in /dev/shm."""
    import multiprocessing as mp
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 3
    procs = [ctx.Process(target=_shared_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [o[1] for o in out] == [1, 0, 0]                      # built once
    assert len({o[2] for o in out}) == 1 and all(o[3] > 0 for o in out)
    assert [o[4] for o in out] == [-7, -8, -9]                   # private writes
    assert not any(o[5] for o in out)                            # file already unlinked


def _two_ranks_in_turn(shared, build, tag):
    """rank 0, then rank 1, in this process: the broadcast is a box rank 0 fills and rank 1 reads"""
    box = {}
    r0 = shared.build_shared(build, 0, lambda x: box.setdefault("w", x), lambda: None, tag)
    return r0, box


def test_shared_generation_falls_back_when_the_file_cannot_be_written(tmp_path, monkeypatch):
    """No room / no /dev/shm: rank 0 broadcasts that there is no file and every other rank builds its own copy -- slower,
    never wrong; a partial file is not left behind.  A failed GENERATION is broadcast too and raises on every rank."""
    import os
    import pytest
    from alfi_amd import shared
    made = []

    def build():
        made.append(1)
        return {"a": np.arange(1000, dtype=np.int64), "b": [np.ones(7), "x"]}
    # a directory that does not exist: the file cannot be created
    monkeypatch.setattr(shared, "_base", lambda: str(tmp_path / "missing"))
    r0, box = _two_ranks_in_turn(shared, build, "t")
    assert box["w"] == ("file", None)
    r1 = shared.build_shared(build, 1, lambda x: box["w"], lambda: None, "t")
    assert len(made) == 2 and np.array_equal(r0["a"], r1["a"])
    # the dump dies half way (disk full): nothing stays behind, rank 1 builds
    monkeypatch.setattr(shared, "_base", lambda: str(tmp_path))
    real_dumps = shared.pickle.dumps
    monkeypatch.setattr(shared.pickle, "dumps", lambda *a, **k: (_ for _ in ()).throw(OSError(28, "No space left on device")))
    made.clear()
    r0, box = _two_ranks_in_turn(shared, build, "u")
    monkeypatch.setattr(shared.pickle, "dumps", real_dumps)
    assert box["w"] == ("file", None) and os.listdir(str(tmp_path)) == []
    r1 = shared.build_shared(build, 1, lambda x: box["w"], lambda: None, "u")
    assert len(made) == 2 and np.array_equal(r0["a"], r1["a"])
    # generation fails on rank 0: rank 0 re-raises its error, the others raise instead of waiting or building
    def bad():
        raise ValueError("no such config")
    box = {}
    with pytest.raises(ValueError):
        shared.build_shared(bad, 0, lambda x: box.setdefault("w", x), lambda: None, "v")
    assert box["w"][0] == "error"
    with pytest.raises(RuntimeError):
        shared.build_shared(build, 1, lambda x: box["w"], lambda: None, "v")
    # and the normal path round-trips through the file: arrays equal, copy-on-write views
    obj = build()
    shared.dump(obj, str(tmp_path / "g_v"))
    back = shared.load(str(tmp_path / "g_v"))
    assert np.array_equal(back["a"], obj["a"]) and back["b"][1] == "x"
    back["a"][0] = -1                                   # private write
    assert shared.load(str(tmp_path / "g_v"))["a"][0] == 0


def test_shared_file_is_private_and_only_a_private_file_is_unpickled(tmp_path, monkeypatch):
    """the file rank 0 writes has an unpredictable name, is created exclusively with mode 0600, and a rank
    unpickles a file only if it is a regular file of its own uid that nobody else can write -- never through a link, never
    a name somebody else may have planted."""
    import os
    import pytest
    import stat
    from alfi_amd import shared
    monkeypatch.setattr(shared, "_base", lambda: str(tmp_path))
    seen = {}

    def bcast(x):
        seen["w"] = x
        st = os.stat(x[1])
        seen["mode"], seen["uid"] = stat.S_IMODE(st.st_mode), st.st_uid
        return x
    obj = {"a": np.arange(10)}
    shared.build_shared(lambda: obj, 0, bcast, lambda: None, "cfg/with:odd chars")
    name = os.path.basename(seen["w"][1])
    assert seen["mode"] == 0o600 and seen["uid"] == os.getuid()
    assert name.startswith("alfi_gen_cfg_with_odd_chars_") and len(name) > len("alfi_gen_cfg_with_odd_chars_.bin") + 6
    assert os.listdir(str(tmp_path)) == []                       # unlinked after the barrier
    # two jobs with the same tag get different files (round 3: one name per uid / MASTER_PORT / config)
    fd1, p1 = shared._create("same")
    fd2, p2 = shared._create("same")
    os.close(fd1), os.close(fd2)
    assert p1 != p2
    # dump never overwrites, never follows a link
    good = str(tmp_path / "good")
    shared.dump(obj, good)
    with pytest.raises(FileExistsError):
        shared.dump(obj, good)
    os.symlink(good, str(tmp_path / "link"))
    with pytest.raises(OSError):
        shared.dump(obj, str(tmp_path / "link"))
    # load refuses a link and a file others can write
    with pytest.raises(OSError):
        shared.load(str(tmp_path / "link"))
    os.chmod(good, 0o666)
    with pytest.raises(PermissionError):
        shared.load(good)
    os.chmod(good, 0o600)
    assert np.array_equal(shared.load(good)["a"], obj["a"])
    # ... and a file of another owner (fstat patched: the test cannot chown)
    real_fstat = os.fstat

    class _St(object):
        def __init__(self, st):
            self.st_mode, self.st_uid = st.st_mode, st.st_uid + 1
    monkeypatch.setattr(shared.os, "fstat", lambda fd: _St(real_fstat(fd)))
    with pytest.raises(PermissionError):
        shared.load(good)


def test_lazy_operator_with_the_full_grad_div_term():
    """LazyOperator(full_div=True): rows of the Scott-Vogelius level operator (gamma (div u, div v), solver.py:616) for a node
    subset equal the rows of the globally assembled one."""
    from alfi_amd.lazy import LazyOperator
    from alfi_amd.problem import TwoDimLidDrivenCavityProblem
    from alfi_amd.sv import build_sv_hierarchy
    lv, _ = build_sv_hierarchy(TwoDimLidDrivenCavityProblem(2), 1, 2, Re=50.0)
    L = lv[-1]
    V = L.V
    wind = TwoDimLidDrivenCavityProblem(2).driver(V.node_coords)
    op = LazyOperator(V, L.A.rowptr, L.A.colidx, V.mesh.cell_geometry(), V.element.reference_tensors(), L.nu, L.gamma, 1.0,
                      np.ascontiguousarray(wind), full_div=True)
    rows = np.arange(1, V.num_nodes, 3)
    _same_bsr(op.select_rows(rows), L.A.select_rows(rows))
