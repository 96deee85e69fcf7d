"""Partitioned HIP multigrid (alfi_amd/dist.py + the library's exchange callbacks) against the single-GPU HIP result and
the oracle.  -m gpu.  The box has one GPU, so 2 or 3 ranks share it and exchange through gloo (host-staged); the library's
pack / callback / unpack sequence is exactly what runs over RCCL on 8 GPUs."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CYCLE_TOL = 1e-5      # see tests/test_gpu_parity.py


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("case,world,robust,overlap", [("2d-all-distributed", 2, 0, "1"), ("2d-coarse-on-rank0", 3, 1, "1"),
                                                       ("3d-P2FB", 2, 0, "1"), ("3d-P1FB", 3, 1, "1"),
                                                       ("3d-P2FB-3lev", 4, 1, "1"), ("3d-P2FB", 2, 1, "0"),
                                                       ("2d-SV", 3, 1, "1"), ("3d-SV-P3", 2, 1, "1")])
def test_partitioned_cycles_match_single_gpu(case, world, robust, overlap, tmp_path):
    _partitioned_cycles(case, world, robust, overlap, tmp_path, {})


@pytest.mark.parametrize("case,world,robust,overlap", [("2d-all-distributed", 2, 0, "1"), ("2d-coarse-on-rank0", 3, 1, "0"),
                                                       ("3d-P2FB", 2, 1, "1"), ("3d-P2FB-3lev", 4, 1, "1"),
                                                       ("3d-SV-P3", 2, 1, "0")])
def test_native_transport_with_several_ranks(case, world, robust, overlap, tmp_path):
    """The PRODUCT transport (csrc/comm.hip: the library's own communicator, neighbour tables, grouped ncclSend / ncclRecv,
    all-reduces, the asynchronous side stream of the overlapped exchanges) with 2-4 ranks.  RCCL refuses several ranks on
    one device, so on this one-GPU box the nine librccl entry points comm.hip resolves are served by
    tests/mock_rccl (ALFI_RCCL_LIB): shared-memory mailboxes between the rank processes, same call sequence, same
    message sizes and ordering -- a wrong neighbour table, count or offset fails here as it would over xGMI."""
    from tests.mock_rccl.build import build
    lib = build()
    _partitioned_cycles(case, world, robust, overlap, tmp_path, {"ALFI_DIST_TRANSPORT": "rccl", "ALFI_RCCL_LIB": lib,
                                                                  "ALFI_TEST_EXPECT_TRANSPORT": "rccl"})


def test_overlap_threshold_between_the_ranks_shares(tmp_path):
    """the overlapped smoother iteration exchanges forward / reverse / forward, the plain one forward / sum.  With
    uneven shares and ALFI_DIST_OVERLAP_MIN_DOFS between them a per-rank decision pairs send/recv groups of different counts
    and directions.  The decision is now collective (smallest share of the level, alfi_amd.dist.overlap_decision): the run
    with the threshold between the two ranks' shares matches the single-GPU result like every other run."""
    from alfi_amd.dist import choose_splits
    from tests.mock_rccl.build import build
    from tests.test_dist_cpu import _hier
    lv, tr, k, min_dofs = _hier("2d-all-distributed")
    shares = np.diff(choose_splits(lv, 2, min_dofs)[-1]) * lv[-1].bs
    assert shares.min() != shares.max(), "the weighted split of this case is expected to be uneven"
    thr = int(shares.min() + shares.max()) // 2
    assert shares.min() < thr <= shares.max()
    _partitioned_cycles("2d-all-distributed", 2, 1, "1", tmp_path,
                        {"ALFI_DIST_TRANSPORT": "rccl", "ALFI_RCCL_LIB": build(), "ALFI_TEST_EXPECT_TRANSPORT": "rccl",
                         "ALFI_DIST_OVERLAP_MIN_DOFS": str(thr)})


@pytest.mark.parametrize("exact", ["1", "0"])
def test_nearly_invariant_krylov_space_on_two_ranks(tmp_path, exact):
    """the partitioned smoother used to take |w - V h| from |w|^2 - |h|^2 (one all-reduce per iteration): absolute
    error eps |w|^2, so a new direction that is small against w -- a Krylov space that is nearly invariant, as late smoothing
    steps and nearly converged outer iterations produce -- is mis-normalised.  The default is now the second reduction
    (PETSc's VecNorm).  Here b lies in the span of three eigenvectors of A M^-1 up to 1e-7: after three iterations the new
    direction is 1e-7 of w.  Two ranks over the mock transport, FGMRES(7): with the default the true residual equals the
    single-GPU smoother's (which always takes the norm from the vector) to 1e-6 relative; the Pythagorean form
    (ALFI_DIST_EXACT_NORM=0, kept for measurements) is only required to stay finite."""
    import scipy.linalg
    from alfi_amd import hip
    from alfi_amd.problem import TwoDimLidDrivenCavityProblem, build_hierarchy
    from oracle import alfi_oracle as O
    from tests.mock_rccl.build import build
    import tests.test_dist_cpu as T
    case = "2d-invariant"
    lv, tr, k, _ = T._hier(case)
    L = lv[-1]
    ol = O.build_oracle_mg(lv, tr, k).levels[-1]
    A = ol["A"].toarray()
    Minv = np.column_stack([ol["smoother"].apply(e) for e in np.eye(L.n)])
    w, V = scipy.linalg.eig(A @ Minv)
    free = np.setdiff1d(np.arange(L.n), L.bc_dofs)
    real = [i for i in np.argsort(-np.abs(w)) if abs(w[i].imag) < 1e-12 and np.abs(V[free, i]).max() > 1e-8][:3]
    b = np.real(V[:, real]).sum(axis=1)
    b /= np.abs(b).max()
    b += 1e-7 * np.random.default_rng(1).standard_normal(L.n)
    b[L.bc_dofs] = 0.0
    bfile = os.path.join(str(tmp_path), "b.npy")
    np.save(bfile, b)
    world, port = 2, _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="4", ALFI_DIST_TRANSPORT="rccl", ALFI_RCCL_LIB=build(),
                   ALFI_DIST_EXACT_NORM=exact)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py"), case, "0",
                                       str(tmp_path), str(k), bfile], env=env, cwd=ROOT))
    ctx = hip.Context(0)
    mg = hip.Multigrid(ctx, lv, tr, k)
    db, dx = ctx.vec(b), ctx.vec(L.n)
    mg.levels[-1].smooth(k, db, dx, nonzero_guess=False)
    xs = dx.get()
    mg.close()
    ctx.close()
    for p in procs:
        assert p.wait(timeout=600) == 0
    xd = np.full_like(b, np.nan)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        xd[z["dofs"]] = z["xs"]
    assert np.isfinite(xd).all()
    rs, rd = np.linalg.norm(b - A @ xs), np.linalg.norm(b - A @ xd)
    # three iterations remove the eigenvector part: what is left is the 1e-7 perturbation
    assert rs < 1e-5 * np.linalg.norm(b)
    if exact == "1":
        assert abs(rd - rs) <= 1e-6 * rs + 1e-14, (rs, rd)
        assert np.abs(xd - xs).max() <= 1e-7 * np.abs(xs).max(), np.abs(xd - xs).max() / np.abs(xs).max()
    else:
        print("Pythagorean norm: true residual %.6e against %.6e (single GPU)" % (rd, rs))


def _partitioned_cycles(case, world, robust, overlap, tmp_path, extra_env):
    from alfi_amd import hip
    from oracle import alfi_oracle as O
    from tests.test_dist_cpu import _hier
    lv, tr, k, _ = _hier(case)
    b = np.random.default_rng(0).standard_normal(lv[-1].n)
    b[lv[-1].bc_dofs] = 0.0
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="4", ALFI_DIST_OVERLAP=overlap,
                   ALFI_DIST_OVERLAP_MIN_DOFS="0")       # exercise the overlapped sequence on these small levels too
        env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py"), case,
                                       str(robust), str(tmp_path)], env=env, cwd=ROOT))
    # the single-GPU references while the ranks run
    ctx = hip.Context(0)
    mg = hip.Multigrid(ctx, lv, tr, k, robust_restriction=bool(robust))
    db, dx = ctx.vec(b), ctx.vec(lv[-1].n)
    mg.vcycle(db, dx)
    mg.vcycle(db, dx)
    sv = dx.get()
    mg.fcycle(db, dx)
    sf = dx.get()
    mg.close()
    ctx.close()
    omg = O.build_oracle_mg(lv, tr, k, schoeberl_restriction=bool(robust))
    top = len(lv) - 1
    ov = omg.vcycle(top, b, omg.vcycle(top, b, np.zeros_like(b)))
    for p in procs:
        assert p.wait(timeout=600) == 0
    dv, df = np.full_like(b, np.nan), np.full_like(b, np.nan)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        dv[z["dofs"]], df[z["dofs"]] = z["xv"], z["xf"]
    assert not np.isnan(dv).any() and not np.isnan(df).any()
    assert np.abs(dv - sv).max() / np.abs(sv).max() < CYCLE_TOL
    assert np.abs(df - sf).max() / np.abs(sf).max() < CYCLE_TOL
    assert np.abs(dv - ov).max() / np.abs(ov).max() < CYCLE_TOL


@pytest.mark.parametrize("exact_norm,tol,transport,sparse_min", [("1", 1e-6, "rccl", None), ("0", CYCLE_TOL, "rccl", None),
                                                                 ("1", 1e-6, "callback", None),
                                                                 # the coarse owner factors with the multifrontal solver
                                                                 # (local numbering, level-set bisection)
                                                                 ("1", 1e-6, "rccl", "0")])
def test_rccl_code_path_with_a_one_rank_group(tmp_path, exact_norm, tol, transport, sparse_min):
    """The RCCL transports of alfi_amd.dist on the one GPU of the box: a 1-rank process group with the exchange points
    forced on (empty halos, 1-rank all-reduces).  "rccl": the library's own communicator (alfi_ctx_comm_init from a unique
    id, the exchanges issued by the library on its stream -- the product path of bench.py --gpus N); "callback": the
    library calls back into alfi_amd.dist, which issues torch.distributed collectives on the library's stream.  Checks
    that RCCL accepts exactly the calls the 8-GPU run makes and that the cycle still matches the
    single-GPU result (exact-norm variant: the same algorithm, but the single-GPU smoother sums its reduction partials
    inside the consuming kernels and multiplies with whole-row SpMV chunks, i.e. in another order -- measured effect of a
    pure change of summation order on this case: 2e-8 .. 6e-8, scripts/ab_cycle.py) (to rounding with the exact-norm variant; with the one-all-reduce-per-iteration default the 1e-14-level difference in
    |w| is amplified by the chained FGMRES least-squares problems like any other rounding difference: CYCLE_TOL)."""
    import textwrap
    script = tmp_path / "one_rank.py"
    script.write_text(textwrap.dedent('''
        import os, sys
        import numpy as np
        sys.path.insert(0, %r)
        import torch, torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        from tests.test_dist_cpu import _hier
        from alfi_amd import hip
        from alfi_amd.dist import DistMultigrid
        lv, tr, k, _ = _hier("3d-P2FB")
        dmg = DistMultigrid(lv, tr, k, robust_restriction=True, min_dofs=1, force_distributed=True)
        assert dmg.comm.backend == "nccl" and all(p.distributed for p in dmg.parts[1:])
        assert dmg.transport == os.environ["ALFI_DIST_TRANSPORT"]
        b = np.random.default_rng(0).standard_normal(lv[-1].n)
        b[lv[-1].bc_dofs] = 0.0
        db, dx = dmg.local_vec(b), dmg.local_vec()
        dmg.vcycle(db, dx); dmg.vcycle(db, dx)
        xv = dmg.owned(dx)
        dmg.fcycle(db, dx)
        xf = dmg.owned(dx)
        comm_ms = dmg.ctx.prof_get()["COMM"]
        mg = hip.Multigrid(dmg.ctx, lv, tr, k, robust_restriction=True)
        sb, sx = dmg.ctx.vec(b), dmg.ctx.vec(lv[-1].n)
        with torch.cuda.stream(dmg.stream):
            mg.vcycle(sb, sx); mg.vcycle(sb, sx)
            sv = sx.get()
            mg.fcycle(sb, sx)
            sf = sx.get()
        assert np.abs(xv - sv).max() <= %g * np.abs(sv).max(), np.abs(xv - sv).max()
        assert np.abs(xf - sf).max() <= %g * np.abs(sf).max()
        dist.destroy_process_group()
        print("ONE-RANK-RCCL-OK")
    ''' % (ROOT, tol, tol)))
    # ALFI_DIST_EXACT_NORM=1: |w| by its own all-reduce (PETSc's VecNorm) -> the partitioned path reproduces the serial one
    # to rounding; default: |w|^2 = |w_old|^2 - |h|^2 from the single all-reduce of the iteration
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), ALFI_DIST_EXACT_NORM=exact_norm,
               ALFI_DIST_TRANSPORT=transport, ALFI_DIST_OVERLAP_MIN_DOFS="0")      # the asynchronous (overlapped) exchanges too, on these small levels
    if sparse_min is not None:
        env["ALFI_COARSE_SPARSE_MIN"] = sparse_min
    out = subprocess.run([sys.executable, str(script)], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ONE-RANK-RCCL-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


@pytest.mark.parametrize("case,world", [("2d-all-distributed", 2), ("3d-P2FB", 3)])
def test_partitioned_multiplicative(case, world, tmp_path):
    """Multiplicative sweeps on partitioned levels (local Gauss-Seidel per rank, additive between ranks -- PCPATCH's MPI
    semantics [3P]): the HIP path against the SPMD oracle on the same rank-local data."""
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="4")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_mult_worker.py"), case,
                                       str(tmp_path)], env=env, cwd=ROOT))
    for p in procs:
        assert p.wait(timeout=600) == 0
    for r in range(world):
        assert os.path.exists(os.path.join(str(tmp_path), "rank%d.ok" % r))


@pytest.mark.parametrize("case,world", [("2d-all-distributed", 2), ("3d-P2FB", 3), ("2d-SV", 2), ("3d-SV-P3", 2)])
def test_partitioned_outer_solve(case, world, tmp_path):
    """alfi_amd.dist.DistSaddle (host-driven FGMRES around the library's partitioned cycles, halo routes and divergence
    products) against the single-GPU alfi_saddle_solve: same iteration count, same solution."""
    from alfi_amd import hip
    from alfi_amd.problem import build_pressure_coupling
    from tests.test_dist_cpu import _hier
    lv, tr, k, _ = _hier(case)
    L = lv[-1]
    b = np.random.default_rng(0).standard_normal(L.n)
    b[L.bc_dofs] = 0.0
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="4")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_saddle_worker.py"), case,
                                       str(tmp_path)], env=env, cwd=ROOT))
    ctx = hip.Context(0)
    mg = hip.Multigrid(ctx, lv, tr, k, robust_restriction=False)
    if "SV" in case:      # Scott-Vogelius pair: discontinuous P_{k-1} pressure, block-diagonal mass inverse (DGMassInv)
        from alfi_amd.sv import build_sv_pressure_coupling
        B, _, Minv = build_sv_pressure_coupling(L)
        sad = hip.Saddle(mg, B, None, L.nu, L.gamma, remove_constant_nullspace=True, mass_inv=Minv)
    else:
        B, vol = build_pressure_coupling(L)
        sad = hip.Saddle(mg, B, vol, L.nu, L.gamma, remove_constant_nullspace=True)
    db, dx = ctx.vec(np.concatenate([b, np.zeros(B.shape[0])])), ctx.vec(L.n + B.shape[0])
    its_s, rn = sad.solve(db, dx, 1e-9, 1e-12, 500, 30)
    xs = dx.get()
    sad.close()
    mg.close()
    ctx.close()
    for p in procs:
        assert p.wait(timeout=600) == 0
    xu, xp = np.full(L.n, np.nan), np.full(B.shape[0], np.nan)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        xu[z["dofs"]], xp[z["cells"]] = z["xu"], z["xp"]
        assert abs(int(z["its"]) - its_s) <= 1
        assert float(z["rn"]) <= 2e-9 * np.linalg.norm(b)
    assert not np.isnan(xu).any() and not np.isnan(xp).any()
    assert np.abs(xu - xs[:L.n]).max() < 1e-6 * np.abs(xs[:L.n]).max()
    assert np.abs(xp - xs[L.n:]).max() < 1e-5 * np.abs(xs[L.n:]).max()


@pytest.mark.parametrize("disc,world,transport", [("pkp0", 2, "callback"), ("pkp0-3d", 3, "callback"), ("sv", 2, "callback"),
                                                  ("pkp0-supg", 2, "callback"), ("pkp0", 3, "rccl"), ("pkp0-3d", 2, "rccl")])
def test_partitioned_newton(tmp_path, disc, world, transport):
    """Newton + Reynolds continuation with every linear solve on partitioned levels: same Newton / Krylov counts and the same
    solution as the single-GPU solver.  The operators are refreshed ON THE DEVICE, every rank its own rows
    (alfi_level_set_assembly on partitioned levels): no host assembly during the Newton loops, values equal to the rank-local
    host assembly to 1e-12.  ``sv``: the Scott-Vogelius pair on the barycentric hierarchy (macro-star patches as condensed
    factors, discontinuous P1 pressure owned cell by cell, block DGMassInv); ``pkp0-supg``: with the SUPG terms of the reference's
    production runs (stabilisation.py:47-97), assembled on the device from the rank's cells.
    The Newton state lives DISTRIBUTED on the devices (nested hierarchies: every rank its owned velocity and pressure dofs; the
    levels' refresh states come from it by one halo exchange and index gathers, alfi_amd.dist.StateExchange): per Newton step
    the library copies less than 1 KB between host and device on every rank (alfi_transfer_stats).  ``rccl``: the product
    transport over tests/mock_rccl (several ranks on one device), ``callback``: gloo."""
    from alfi_amd.nssolver import HipNavierStokesSolver, run_solver
    from alfi_amd.problem import TwoDimLidDrivenCavityProblem, ThreeDimLidDrivenCavityProblem
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="4")
        if transport == "rccl":
            from tests.mock_rccl.build import build
            env.update(ALFI_DIST_TRANSPORT="rccl", ALFI_RCCL_LIB=build())
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_newton_worker.py"),
                                       str(tmp_path), disc], env=env, cwd=ROOT))
    s = (HipNavierStokesSolver(TwoDimLidDrivenCavityProblem(4), 2, 2, discretisation="sv") if disc == "sv"
         else HipNavierStokesSolver(ThreeDimLidDrivenCavityProblem(2), 1, 2) if disc == "pkp0-3d"
         else HipNavierStokesSolver(TwoDimLidDrivenCavityProblem(8), 1, 2, stabilisation_type="supg") if disc == "pkp0-supg"
         else HipNavierStokesSolver(TwoDimLidDrivenCavityProblem(8), 1, 2))
    res = run_solver(s, [10, 100])
    for p in procs:
        assert p.wait(timeout=600) == 0
    z = np.load(os.path.join(str(tmp_path), "newton.npz"))
    assert all(z["device_assembly"]) and list(z["host_assemblies"]) == [0] * world, (z["device_assembly"], z["host_assemblies"])
    if disc != "sv":
        assert all(z["resident"]) and max(z["bytes_per_step"]) <= 1024, (z["resident"], z["bytes_per_step"])
    assert max(z["asm_err"]) < 1e-12, z["asm_err"]
    assert all(z["conv"]) and all(res[r]["converged"] for r in (10, 100))
    assert list(z["newton"]) == [res[r]["nonlinear_iter"] for r in (10, 100)]
    assert all(abs(int(a) - res[r]["linear_iter"]) <= 2 for a, r in zip(z["its"], (10, 100)))
    assert np.abs(z["u"] - s.u).max() < 1e-7 * np.abs(s.u).max()
    assert np.abs(z["p"] - s.p).max() < 1e-6 * np.abs(s.p).max()
    s.close()


def _bench(args, env_extra, timeout=900):
    import json
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)                                     # the launcher under test sets them itself
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, capture_output=True,
                         text=True, timeout=timeout)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return out, (json.loads(lines[-1]) if lines else None)


def test_bench_starts_its_own_ranks_shared_gpu():
    """`python bench.py --gpus N` without a launcher starts the N rank processes itself (before anything touches a GPU),
    relays rank 0's JSON line and reports the process group's size.  Here: 2 ranks sharing the box's GPU over gloo (callback
    transport, rank-local generation) -- a functional run of exactly the driver's command shape."""
    out, d = _bench(["--gpus", "2", "--config", "tiny", "--steps", "2", "--warmup", "1"],
                    {"ALFI_DIST_BACKEND": "gloo", "ALFI_DIST_MIN_DOFS": "1000"})
    assert out.returncode == 0 and d is not None, out.stderr[-3000:]
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["config"]["transport"] == "callback"
    assert "shared through /dev/shm" in d["config"]["generation"] and "values rank-local" in d["config"]["generation"]
    assert d["rel_residual_after_timed_cycles"] < 0.5
    assert len(d["setup_s"]["host_peak_rss_GB_per_rank"]) == 2


def test_bench_multi_rank_with_the_native_transport():
    """The driver's multi-GPU bench command with the product transport (library-owned communicator, no Python at the
    exchange points, rank-local generation) on 4 ranks; librccl's entry points served by tests/mock_rccl because the box
    has one GPU.  What the 8-GPU run executes, minus RCCL's own wire protocol."""
    from tests.mock_rccl.build import build
    out, d = _bench(["--gpus", "4", "--config", "tiny", "--steps", "2", "--warmup", "1"],
                    {"ALFI_DIST_BACKEND": "gloo", "ALFI_DIST_MIN_DOFS": "500", "ALFI_DIST_TRANSPORT": "rccl",
                     "ALFI_RCCL_LIB": build()})
    assert out.returncode == 0 and d is not None, out.stderr[-3000:]
    assert d["n_gpus"] == 4 and d["n_ranks_seen"] == 4 and d["config"]["transport"] == "rccl"
    assert d["n_ranks_seen_source"].startswith("alfi_ctx_comm_size") and d["process_group_size"] == 4
    assert 1 <= d["neighbours_max"] <= 3 and d["single_owner_levels_ms"] > 0.0
    assert set(d["per_rank"]["setup_s"]) == {"comm_init", "partition", "localize", "upload_factor"}
    assert d["config"]["robust_restriction"] is False
    assert d["rel_residual_after_timed_cycles"] < 0.5
    # (the smoother's merged reverse-add + forward exchange, alfi_level_set_sum_exchange: two halo exchanges per FGMRES
    # iteration; the three-exchange sequence stays on the callback transport, tests above)
    assert d["per_rank"]["halo_exchanges_per_cycle"][0] > 0


@pytest.mark.parametrize("rank_watchdog", ["1", "0"])
def test_bench_watchdog_ends_a_hung_run(rank_watchdog):
    """A rank that never reaches the first collective of the cycle (fault injection: ALFI_BENCH_TEST_HANG=<rank>:<stage>)
    leaves its peers blocked in an exchange.  `bench.py --gpus 2` must then exit non-zero within ALFI_BENCH_TIMEOUT_S (plus the
    launcher's grace period), end the children it started and say which stage every rank was in -- once with the ranks' own
    watchdog threads (the torch.distributed.run launch has only those), once with the launcher's alone."""
    import time
    from tests.mock_rccl.build import build
    t0 = time.time()
    out, d = _bench(["--gpus", "2", "--config", "tiny", "--steps", "1", "--warmup", "1"],
                    {"ALFI_DIST_BACKEND": "gloo", "ALFI_DIST_MIN_DOFS": "1000", "ALFI_DIST_TRANSPORT": "rccl",
                     "ALFI_RCCL_LIB": build(), "ALFI_BENCH_TEST_HANG": "1:first_cycle", "ALFI_BENCH_TIMEOUT_S": "16",
                     "ALFI_BENCH_GRACE_S": "3", "ALFI_BENCH_RANK_WATCHDOG": rank_watchdog}, timeout=240)
    took = time.time() - t0
    assert out.returncode != 0 and d is None, out.stdout[-500:]
    assert took < 120, took
    assert "WATCHDOG" in out.stderr and "first_cycle" in out.stderr, out.stderr[-3000:]
    assert "rank 1" in out.stderr


def test_bench_launcher_fails_loudly_when_a_rank_cannot_start():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer GPUs than ranks")
    out, d = _bench(["--gpus", "2", "--config", "tiny", "--steps", "1", "--warmup", "0"], {"ALFI_DIST_BACKEND": "nccl"}, 300)
    assert out.returncode != 0 and d is None


def test_two_rank_rccl_run():
    """The native RCCL transport with more than one rank: needs two GPUs (RCCL refuses several ranks on one device)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("fewer than 2 GPUs visible")
    out, d = _bench(["--gpus", "2", "--config", "cfg4t", "--steps", "3", "--warmup", "1"], {"ALFI_DIST_MIN_DOFS": "1000"})
    assert out.returncode == 0 and d is not None, out.stderr[-3000:]
    assert d["n_ranks_seen"] == 2 and d["config"]["transport"] == "rccl" and d["config"]["backend"] == "nccl"
    assert d["rel_residual_after_timed_cycles"] < 0.5


def test_real_rccl_send_recv_to_self(tmp_path):
    """The REAL point-to-point path of the native transport with NON-EMPTY buffers on the one GPU of the box: a 1-rank RCCL
    communicator whose rank names itself as its neighbour (test hook alfi_ctx_comm_allow_self; RCCL accepts a grouped
    ncclSend + ncclRecv to one's own rank).  alfi_level_halo_forward, alfi_level_halo_reverse_add and the merged sum exchange
    (pack kernel, ONE group of send / receive pairs on the library's stream, unpack / add kernels: csrc/comm.hip:native_exchange)
    against the same index arithmetic in NumPy.  Reference: PETSc's VecScatter forward / reverse-add around PCApply_PATCH
    and MatMult under alfi/solver.py:604-605, alfi/relaxation.py:120-121."""
    import textwrap
    script = tmp_path / "self_exchange.py"
    script.write_text(textwrap.dedent('''
        import sys
        import numpy as np
        sys.path.insert(0, %r)
        from alfi_amd import hip, _lib
        from alfi_amd.problem import BSR
        rng = np.random.default_rng(7)
        bs, n_own, n_ghost = 3, 5000, 1300
        nb = n_own + n_ghost
        A = BSR(nb, nb, bs, np.arange(nb + 1, dtype=np.int32), np.arange(nb, dtype=np.int32), np.tile(np.eye(bs), (nb, 1, 1)))
        ctx = hip.Context(0)
        ctx.comm_init(_lib.comm_unique_id(), 0, 1)
        assert ctx.comm_size() == (0, 1)
        ctx.comm_allow_self(True)
        L = hip.Level(ctx, A, np.zeros(0, dtype=np.int32))
        # ghost slot j is a copy of owned node send_nodes[j] (several ghosts may copy one node)
        send_nodes = rng.integers(0, n_own, n_ghost).astype(np.int32)
        L.set_partition(n_own, True, send_nodes, None, None, n_ghost)
        L.set_neighbours([0], [n_ghost], [n_ghost])
        v = rng.standard_normal(nb * bs)
        V = v.reshape(nb, bs)
        # forward: owner -> ghost copies
        dv = ctx.vec(v)
        ctx.comm_stats(reset=True)
        L.halo_forward(dv)
        got = dv.get().reshape(nb, bs)
        want = V.copy(); want[n_own:] = V[send_nodes]
        assert np.array_equal(got, want), "forward"
        # reverse-add: ghost values summed onto their owners, in buffer order
        dv = ctx.vec(v)
        L.halo_reverse_add(dv)
        got = dv.get().reshape(nb, bs)
        want = V.copy()
        for j in np.argsort(send_nodes, kind="stable"):
            want[send_nodes[j]] += V[n_own + j]
        assert np.abs(got - want).max() <= 1e-15 * np.abs(want).max() and np.array_equal(got[n_own:], V[n_own:]), "reverse-add"
        # merged sum exchange: every shared node = its own value + the values received for it, in list order
        ns = 900
        sum_send = rng.integers(0, nb, ns).astype(np.int32)              # what this rank sends (and, to itself, receives)
        shared = np.sort(rng.choice(nb, 400, replace=False)).astype(np.int32)
        ptr, src = [0], []
        for u in shared:
            k = int(rng.integers(1, 4))
            lst = list(rng.integers(0, ns, k))
            lst.insert(int(rng.integers(0, k + 1)), -1)                    # exactly one own value, at a random place
            src += lst; ptr.append(len(src))
        L.set_sum_exchange([0], [ns], sum_send, shared, np.asarray(ptr, dtype=np.int32), np.asarray(src, dtype=np.int32))
        dv = ctx.vec(v)
        L.halo_sum(dv)
        got = dv.get().reshape(nb, bs)
        want = V.copy()
        for i, u in enumerate(shared):
            acc = np.zeros(bs)
            for s_ in src[ptr[i]:ptr[i + 1]]:
                acc = acc + (V[u] if s_ < 0 else V[sum_send[s_]])
            want[u] = acc
        assert np.abs(got - want).max() <= 1e-15 * np.abs(want).max(), "sum exchange"
        halos, reds, sent = ctx.comm_stats()
        assert halos == 3 and sent == (n_ghost + n_ghost + ns) * bs, (halos, sent)
        print("SELF-EXCHANGE-OK")
    ''' % (ROOT,)))
    out = subprocess.run([sys.executable, str(script)], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "SELF-EXCHANGE-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_solo_rank_mode_runs_every_rank_of_a_partition_alone():
    """DistMultigrid(solo=(rank, world)): one rank of a partitioned job alone in the process, exchange points stubbed (SoloComm) --
    the measurement mode of scripts/solo_rank_time.py.  Every rank sets up and cycles, the owned dofs of the ranks add up to the
    level sizes, the library's events see device time in the rank's kernels."""
    from alfi_amd.dist import DistMultigrid
    from alfi_amd.problem import TwoDimLidDrivenCavityProblem, build_hierarchy
    lv, tr = build_hierarchy(TwoDimLidDrivenCavityProblem(8), 2, 2, Re=10.0, lazy=True)
    world = 3
    owned = np.zeros(len(lv), dtype=np.int64)
    for r in range(world):
        dmg = DistMultigrid(lv, tr, 3, solo=(r, world), min_dofs=200)
        b = np.random.default_rng(r).standard_normal(lv[-1].n)
        db, dx = dmg.local_vec(b), dmg.local_vec()
        dmg.ctx.prof_enable(True)
        dmg.ctx.prof_reset()
        dmg.vcycle(db, dx)
        dmg.sync()
        prof = dmg.ctx.prof_get()
        assert prof["PATCH_APPLY"][0] > 0.0 and prof["MATMULT"][0] > 0.0
        assert np.all(np.isfinite(dmg.owned(dx)))
        owned += np.array([int(p.nb_own) * p.bs for p in dmg.parts])
        dmg.close()
    assert np.array_equal(owned, np.array([L.n for L in lv]))
