"""Multi-rank path on CPU (gloo, world size 2 and 3): the product's partitioner / halo plans / exchange
(alfi_amd/dist.py) driven by the SPMD NumPy oracle (oracle/dist_oracle.py) must reproduce the serial oracle's cycles.

The HIP library performs the same exchange sequence through the same Comm object (tests/test_gpu_dist.py, -m gpu)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {
    # name: (dim, baseN, nref, element k, Re, smoother k, min_dofs)
    "2d-all-distributed": (2, 4, 2, 2, 100.0, 3, 1),
    "2d-coarse-on-rank0": (2, 4, 2, 2, 100.0, 3, 500),
    "3d-P2FB": (3, 2, 1, 2, 1000.0, 2, 1),
    "3d-P1FB": (3, 2, 1, 1, 100.0, 2, 1),
    # three 3-D levels, the middle one owned by rank 0 alone, the finest split (the shape of the config-4 run)
    "3d-P2FB-3lev": (3, 1, 2, 2, 100.0, 2, 3000),
    # Scott-Vogelius: macro-star patches (wider ghost layer: solver.py:661-662 asks for overlap 2), macro-cell transfer blocks
    "2d-SV": ("sv2", 2, 2, 2, 100.0, 3, 1),
    "3d-SV-P3": ("sv3", 1, 1, 3, 100.0, 2, 1),
    # a small 2-D level for the nearly-invariant-Krylov-space test of the partitioned smoother (tests/test_gpu_dist.py)
    "2d-invariant": (2, 2, 1, 2, 10.0, 7, 1),
}


def _hier(case):
    from alfi_amd.problem import TwoDimLidDrivenCavityProblem, ThreeDimLidDrivenCavityProblem, build_hierarchy
    dim, baseN, nref, ke, Re, k, min_dofs = CASES[case]
    if dim in ("sv2", "sv3"):
        from alfi_amd.sv import build_sv_hierarchy
        prob = TwoDimLidDrivenCavityProblem(baseN) if dim == "sv2" else ThreeDimLidDrivenCavityProblem(baseN)
        lv, tr = build_sv_hierarchy(prob, nref, ke, Re=Re)
        return lv, tr, k, min_dofs
    prob = TwoDimLidDrivenCavityProblem(baseN) if dim == 2 else ThreeDimLidDrivenCavityProblem(baseN)
    lv, tr = build_hierarchy(prob, nref, ke, Re=Re)
    return lv, tr, k, min_dofs


def _worker(rank, world, port, case, robust, q):
    os.environ["OMP_NUM_THREADS"] = "1"          # up to 8 ranks on the container's 8 cores
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from alfi_amd import dist as D
        from oracle.dist_oracle import DistOracle
        lv, tr, k, min_dofs = _hier(case)
        comm = D.Comm()
        splits = D.choose_splits(lv, world, min_dofs)
        parts = D.build_parts(lv, tr, splits, rank, comm.all_gather_object)
        llev, ltr, lmin = D.localize(lv, tr, parts)
        mg = DistOracle(llev, ltr, lmin, k, comm, robust=robust)
        F = llev[-1]
        p, bs = F.part, F.bs
        b = np.random.default_rng(0).standard_normal(lv[-1].n)
        b[lv[-1].bc_dofs] = 0.0
        bl = np.zeros(F.n)
        bl[:F.n_own] = b[p.own_dofs()]
        xv = mg.vcycle(len(llev) - 1, bl, np.zeros(F.n))
        xv = mg.vcycle(len(llev) - 1, bl, xv)
        xf = mg.fcycle(bl)
        q.put((rank, p.own_dofs(), xv[:F.n_own], xf[:F.n_own]))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("case,world,robust", [("2d-all-distributed", 2, False), ("2d-all-distributed", 3, True),
                                               ("2d-coarse-on-rank0", 2, True), ("3d-P2FB", 2, False),
                                               ("3d-P1FB", 2, True), ("3d-P2FB-3lev", 4, True),
                                               ("2d-SV", 3, True), ("3d-SV-P3", 2, True),
                                               # the driver's widest launch: finest level split eight ways, the rest on rank 0
                                               ("3d-P2FB-3lev", 8, False)])
def test_spmd_oracle_matches_serial(case, world, robust):
    import torch.multiprocessing as mp
    from oracle import alfi_oracle as O
    lv, tr, k, _ = _hier(case)
    ser = O.build_oracle_mg(lv, tr, k, schoeberl_restriction=robust)
    b = np.random.default_rng(0).standard_normal(lv[-1].n)
    b[lv[-1].bc_dofs] = 0.0
    top = len(lv) - 1
    xv = ser.vcycle(top, b, np.zeros_like(b))
    xv = ser.vcycle(top, b, xv)
    xf = ser.fcycle(b)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, robust, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    dv, df = np.full_like(b, np.nan), np.full_like(b, np.nan)
    for rank, dofs, v, f in got:
        dv[dofs], df[dofs] = v, f
    assert not np.isnan(dv).any()
    # 2-D: 1e-8.  3-D (patch condition numbers ~1e7 at Re 1000): LAPACK inverts the same patch operators with the dofs in
    # local instead of global order, a cond*eps ~ 1e-9 difference per apply that the chained FGMRES least-squares problems
    # amplify exactly as in tests/test_gpu_parity.py (CYCLE_TOL 1e-5); a wrong halo or a missing ghost shows up at 1e-2.
    # (2-D Scott-Vogelius: macro-star patches of 50-90 dofs; the host generator sums the cell contributions with OpenMP atomics,
    # so the operator's last bits change from run to run -- 1.06e-8 has been seen under load)
    tol = 1e-7 if case == "2d-SV" else 1e-8 if case.startswith("2d") else 1e-5
    assert np.abs(dv - xv).max() / np.abs(xv).max() < tol
    assert np.abs(df - xf).max() / np.abs(xf).max() < tol


def test_partition_covers_and_plans_are_consistent():
    """Single process: every rank's plan computed locally; owners' send lists mirror the ghosts' receive lists."""
    from alfi_amd import dist as D
    lv, tr, k, _ = _hier("3d-P2FB")
    world = 3
    splits = D.choose_splits(lv, world, 1)
    allparts = [D.build_parts(lv, tr, splits, r, None) for r in range(world)]
    for l in range(len(lv)):
        owned = np.concatenate([np.arange(allparts[r][l].lo, allparts[r][l].hi) for r in range(world)])
        assert np.array_equal(owned, np.arange(lv[l].A.nbrows))
        for r in range(world):
            pr = allparts[r][l]
            gb = pr.ghosts_by_owner()
            for q in range(world):
                pq = allparts[q][l]
                assert np.array_equal(gb[q], pq.own_nodes[pq.send_nodes[r]])
            # owned rows see all their columns locally; owned patches are complete
            LL = D.localize_level(lv[l], pr)
            assert LL.A.nbrows == pr.nb_loc and LL.A.colidx.max(initial=0) < pr.nb_loc
            if l > 0:
                assert sum(len(D.owned_patches(lv[l], allparts[q][l].lo, allparts[q][l].hi)) for q in range(world)) \
                    == len(lv[l].patch_ptr) - 1


def _saddle_worker(rank, world, port, case, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from alfi_amd import dist as D
        from alfi_amd.problem import build_pressure_coupling
        from oracle.dist_oracle import DistOracle, dist_saddle_solve
        lv, tr, k, min_dofs = _hier(case)
        L = lv[-1]
        comm = D.Comm()
        splits = D.choose_splits(lv, world, min_dofs)
        parts = D.build_parts(lv, tr, splits, rank, comm.all_gather_object)
        llev, ltr, lmin = D.localize(lv, tr, parts)
        mg = DistOracle(llev, ltr, lmin, k, comm, robust=False)
        B, vol = build_pressure_coupling(L)
        p = llev[-1].part
        cells, Bloc, md, _ = D.localize_pressure(B, vol, L.V.cell_nodes, p, L.bs)
        b = np.random.default_rng(0).standard_normal(L.n)
        b[L.bc_dofs] = 0.0
        rhs = np.concatenate([b[p.own_dofs()], np.zeros(len(cells))])
        x, its, rn = dist_saddle_solve(mg, Bloc, md, B.shape[0], L.nu, L.gamma, rhs, rtol=1e-9, atol=1e-12)
        q.put((rank, p.own_dofs(), cells, x[:llev[-1].n_own], x[llev[-1].n_own:], its))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case,world", [("2d-all-distributed", 2), ("3d-P2FB", 3)])
def test_spmd_outer_solve_matches_serial(case, world):
    """The outer FGMRES / fieldsplit-Schur solve on partitioned levels (velocity dofs owned with their nodes, pressure dofs
    with their cells): same iteration count and solution as the serial oracle."""
    import torch.multiprocessing as mp
    from alfi_amd.problem import build_pressure_coupling
    from oracle import alfi_oracle as O
    lv, tr, k, _ = _hier(case)
    L = lv[-1]
    ser = O.build_oracle_mg(lv, tr, k, schoeberl_restriction=False)
    B, vol = build_pressure_coupling(L)
    b = np.random.default_rng(0).standard_normal(L.n)
    b[L.bc_dofs] = 0.0
    xs, its_s, _ = O.saddle_solve(ser, ser.levels[-1]["A"], B, vol, L.nu, L.gamma, np.concatenate([b, np.zeros(B.shape[0])]),
                                  rtol=1e-9, atol=1e-12)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_saddle_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    xu, xp = np.full(L.n, np.nan), np.full(B.shape[0], np.nan)
    for rank, dofs, cells, u, pp, its in got:
        xu[dofs], xp[cells] = u, pp
        assert abs(its - its_s) <= 1
    assert not np.isnan(xu).any() and not np.isnan(xp).any()             # every velocity dof and every cell has one owner
    assert np.abs(xu - xs[:L.n]).max() < 1e-6 * np.abs(xs[:L.n]).max()
    assert np.abs(xp - xs[L.n:]).max() < 1e-5 * np.abs(xs[L.n:]).max()


@pytest.mark.parametrize("case,world", [("3d-P2FB-3lev", 4), ("2d-all-distributed", 3), ("3d-P1FB", 2)])
def test_neighbour_tables_of_the_native_transport_are_consistent(case, world):
    """What alfi_level_set_neighbours receives on every rank (and the grouped ncclSend / ncclRecv of comm.hip then use): for
    every pair of ranks the sender's count equals the receiver's, the sender's segment holds exactly the nodes the receiver
    lists as its ghosts owned by the sender, in the receiver's ghost order; the counts add up to the halo sizes; a rank is
    never its own neighbour.  No GPU, no process group: all ranks' partitions are computed here."""
    from alfi_amd import dist as D
    lv, tr, k, min_dofs = _hier(case)
    splits = D.choose_splits(lv, world, min_dofs)
    parts = [D.build_parts(lv, tr, splits, r) for r in range(world)]
    for l in range(len(lv)):
        for a in range(world):
            pa = parts[a][l]
            nbr = np.flatnonzero((pa.send_counts > 0) | (pa.recv_counts > 0))
            assert a not in nbr
            assert pa.send_counts[nbr].sum() == sum(len(s) for s in pa.send_nodes) and pa.recv_counts[nbr].sum() == pa.nb_ghost
            send_off = np.concatenate([[0], np.cumsum(pa.send_counts)])
            send_all = np.concatenate(pa.send_nodes) if pa.send_counts.sum() else np.zeros(0, dtype=np.int64)
            for b in range(world):
                pb = parts[b][l]
                assert pa.send_counts[b] == pb.recv_counts[a]
                if pa.send_counts[b] == 0:
                    continue
                # global ids of what a packs for b == b's ghosts owned by a, in b's ghost order
                seg = pa.own_nodes[send_all[send_off[b]:send_off[b + 1]]]
                roff = np.concatenate([[0], np.cumsum(pb.recv_counts)])
                assert np.array_equal(seg, pb.ghosts[roff[a]:roff[a + 1]])


@pytest.mark.parametrize("case,world", [("3d-P2FB", 4), ("2d-all-distributed", 3)])
def test_sum_exchange_plan(case, world):
    """The plan of the merged reverse-add + forward exchange (LevelPart.set_sum_plan -> alfi_level_set_sum_exchange),
    emulated with NumPy: after ONE exchange among the holders of every shared node, owner and ghost copies all carry the
    sum of the holders' partial values; the neighbour segments of two ranks mirror each other."""
    from alfi_amd.dist import choose_splits, build_parts
    lv, tr, k, _ = _hier(case)
    splits = choose_splits(lv, world, 1)
    parts = [build_parts(lv, tr, splits, r, None) for r in range(world)]
    l = len(lv) - 1
    nb = lv[l].A.nbrows
    rng = np.random.default_rng(0)
    vals = [rng.standard_normal(parts[r][l].nb_loc) for r in range(world)]
    total = np.zeros(nb)
    for r in range(world):
        np.add.at(total, parts[r][l].nodes, vals[r])
    for r in range(world):
        P = parts[r][l]
        bufs = []
        for q, c in zip(P.sum_ranks, P.sum_counts):
            Q = parts[q][l]
            i = list(Q.sum_ranks).index(r)
            o = int(Q.sum_counts[:i].sum())
            assert int(Q.sum_counts[i]) == int(c)
            seg = Q.sum_send_nodes[o:o + int(c)]
            mine = P.sum_send_nodes[int(P.sum_counts[:list(P.sum_ranks).index(q)].sum()):][:int(c)]
            assert np.array_equal(Q.nodes[seg], P.nodes[mine])          # the same global nodes, in the same order
            bufs.append(vals[q][seg])
        recv = np.concatenate(bufs) if bufs else np.zeros(0)
        out = vals[r].copy()
        for i, n in enumerate(P.sum_nodes):
            src = P.sum_src[P.sum_ptr[i]:P.sum_ptr[i + 1]]
            assert np.count_nonzero(src < 0) == 1
            out[n] = sum(vals[r][n] if s < 0 else recv[s] for s in src)
        shared = np.zeros(P.nb_loc, dtype=bool)
        shared[P.sum_nodes] = True
        assert shared[P.nb_own:].all()                                   # every ghost copy takes part
        assert np.abs(out[shared] - total[P.nodes][shared]).max() < 1e-12
        assert np.array_equal(out[~shared], vals[r][~shared])


def test_overlap_decision_is_collective():
    """Every rank must take the same overlap decision for a level (the overlapped and the plain smoother iteration issue
    different exchange sequences): it follows from the split points alone, by the smallest share."""
    from alfi_amd.dist import overlap_decision
    splits = np.array([0, 1000, 2500, 3000])
    for thr, want in ((0, True), (1500, True), (1501, False), (4000, False), (1 << 62, False)):
        got = {overlap_decision(splits, 3, True, True, thr) for _rank in range(3)}
        assert got == {want}, (thr, got)
    assert not overlap_decision(splits, 3, False, True, 0) and not overlap_decision(splits, 3, True, False, 0)
    # a rank with an empty share (single-owner level forced distributed) does not veto
    assert overlap_decision(np.array([0, 1000, 1000]), 2, True, True, 2000)


def test_pressure_rows_of_the_dg_pair_are_owned_cell_by_cell():
    """localize_pressure for the Scott-Vogelius pair: npc pressure dofs per cell (rows c * npc .. of B) go with the cell to the
    owner of its lowest-numbered node; over the ranks every pressure row is owned exactly once, the local B reproduces the
    global rows, and the selected block of the mass inverse is that of the owned cells."""
    sys.path.insert(0, ROOT)
    import scipy.sparse as sp
    from alfi_amd import dist as D
    from alfi_amd.sv import build_sv_pressure_coupling
    lv, tr, _, min_dofs = _hier("2d-SV")
    L = lv[-1]
    B, M, Minv = build_sv_pressure_coupling(L)
    world = 3
    splits = D.choose_splits(lv, world, min_dofs)
    seen = np.zeros(B.shape[0], dtype=int)
    u = np.random.default_rng(1).standard_normal(L.n)
    for r in range(world):
        part = D.build_parts(lv, tr, splits, r, None)[-1]
        prows, Bloc, md, Mi = D.localize_pressure(B, None, L.V.cell_nodes, part, L.bs, Minv)
        assert md is None and len(prows) % 3 == 0                    # P1dg in 2-D: 3 dofs per cell
        seen[prows] += 1
        loc = np.concatenate([part.own_dofs(), (part.ghosts[:, None] * L.bs + np.arange(L.bs)).ravel()])
        assert np.allclose(Bloc @ u[loc], (B @ u)[prows])
        assert abs(Mi - sp.csr_matrix(Minv)[prows][:, prows]).max() == 0.0
        assert np.allclose((Mi @ (sp.csr_matrix(M)[prows][:, prows])).toarray(), np.eye(len(prows)), atol=1e-10)
    assert (seen == 1).all()


@pytest.mark.parametrize("case,world", [("2d-all-distributed", 2), ("3d-P2FB", 3)])
def test_assembly_cells_and_contributor_lists_of_a_partition(case, world):
    """Device-side operator refresh on partitioned levels (alfi/solver.py:320, 325 under :604-605): the cells a rank hands to
    alfi_level_set_assembly touch a local node, their node numbering puts the local nodes first, and the contributor lists of
    the rank's LOCAL operator (columns numbered owned-first, i.e. unsorted rows; ghost rows restricted to local columns) hold
    exactly the (cell, a, b) pairs whose two nodes are row and column of the block -- checked pair by pair."""
    from alfi_amd import _hostlib
    from alfi_amd import dist as D
    lv, tr, k, min_dofs = _hier(case)
    splits = D.choose_splits(lv, world, min_dofs)
    for rank in range(world):
        parts = D.build_parts(lv, tr, splits, rank)
        L, p = lv[-1], parts[-1]
        A = D.localize_operator(L.A, p)
        cells, cn, nodes = D.assembly_cells(L.V, p)
        gcn = np.asarray(L.V.cell_nodes)
        assert np.array_equal(nodes[:p.nb_loc], p.nodes) and np.array_equal(nodes[cn], gcn[cells])
        touched = np.isin(gcn, p.nodes).any(axis=1)
        assert np.array_equal(np.flatnonzero(touched), cells)
        cptr, ccell, cba = _hostlib.contributors(cn, p.nb_loc, A.rowptr, A.colidx, nindex=len(nodes), partial=True)
        nloc = cn.shape[1]
        rows = np.repeat(np.arange(A.nbrows), np.diff(A.rowptr))
        blk = np.repeat(np.arange(len(A.colidx)), np.diff(cptr))
        a, b = cba % nloc, cba // nloc
        assert np.array_equal(cn[ccell, a], rows[blk]) and np.array_equal(cn[ccell, b], A.colidx[blk])
        # ... and none is missing: every pair of local nodes of a cell is a block of the local operator or (ghost rows only) a
        # coupling the rank does not keep
        have = set(zip(ccell.tolist(), cba.tolist()))
        key = {(int(r), int(c)): i for i, (r, c) in enumerate(zip(rows, A.colidx))}
        for c in range(0, cn.shape[0], max(1, cn.shape[0] // 200)):
            for ia in range(nloc):
                for ib in range(nloc):
                    r_, c_ = int(cn[c, ia]), int(cn[c, ib])
                    if (r_, c_) in key:
                        assert (c, ib * nloc + ia) in have
                    else:
                        assert r_ >= p.nb_own or c_ >= p.nb_loc or r_ >= p.nb_loc
        assert (np.diff(cptr) > 0).all()


def test_rank_zero_gets_a_smaller_share_for_its_single_owner_levels():
    """choose_splits: rank 0 owns the levels below ``min_dofs`` alone, so its share of every partitioned level is smaller by that work
    (single-owner bytes x SINGLE_OWNER_TIME_FACTOR; more so for full cycles, which visit the low levels more often), never below
    half an equal share; the other ranks share the rest equally (by bytes per smoother iteration); every node has exactly one owner."""
    from alfi_amd.dist import SINGLE_OWNER_TIME_FACTOR, choose_splits, level_weights
    from alfi_amd.problem import ThreeDimLidDrivenCavityProblem, build_hierarchy
    lv, _ = build_hierarchy(ThreeDimLidDrivenCavityProblem(2), 2, 2, Re=10.0, lazy=True)
    world, min_dofs = 4, 5000
    single = [L.level == 0 or L.n < min_dofs for L in lv]
    assert any(single) and not all(single)
    w = [float(level_weights(L).sum()) for L in lv]
    shares = {}
    for full in (False, True):
        splits = choose_splits(lv, world, min_dofs, full_cycle=full)
        equal = choose_splits(lv, world, min_dofs, balance_single_owner=False)
        visits = [(len(lv) - L.level) if full else 1 for L in lv]
        ws = SINGLE_OWNER_TIME_FACTOR * sum(v * x for v, x, s in zip(visits, w, single) if s)
        wd = sum(v * x for v, x, s in zip(visits, w, single) if not s)
        f0 = max(0.5 / world, (wd + ws) / (world * wd) - ws / wd)
        for L, s, e, one in zip(lv, splits, equal, single):
            assert s[0] == 0 and s[-1] == L.A.nbrows and np.all(np.diff(s) >= 0)
            if one:
                assert np.array_equal(s, e) and s[1] == L.A.nbrows
                continue
            c = np.concatenate([[0.0], np.cumsum(level_weights(L))])
            sh = np.diff(c[s]) / c[-1]
            assert abs(sh[0] - f0) < 2e-3 and sh[0] < 1.0 / world
            assert np.abs(sh[1:] - (1.0 - f0) / (world - 1)).max() < 2e-3
            shares[full] = sh[0]
    assert shares[True] <= shares[False]
