#!/usr/bin/env python3
"""Benchmark of the multigrid hot path: V-cycles/s (and DoF*smooths/s) on the synthetic lid-driven cavity.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg4]

A "step" is one multiplicative V-cycle (k pre- and post-smoothing FGMRES iterations with the additive vertex-star
patch smoother on every level, Schoeberl prolongation, dense coarse solve) on the finest level, with all operators,
patch inverses and vectors resident in HBM before the timed region starts.  Default workload = BASELINE.json config 4:
ldc3d [P2+FB]^3-P0, N = 56 (10 707 315 velocity dofs, 185 193 star patches of up to 153 dofs), Re = 1000, k = 10.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (patch_apply_kernel: streams every dense patch
inverse once per smoother iteration); its `achieved` is algorithmic bytes / device time measured live with HIP events on
the library's stream during the timed region.  `cpu_baseline` times the C/OpenMP oracle port on the host cores on a
bounded sample (the same hierarchy truncated by one level, scaled by the dof ratio).
"""
import argparse
import json
import os
import resource
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (dim, baseN, nref, k_element, Re, smoothing k)  -- BASELINE.md table
    "cfg1": (2, 16, 2, 2, 10.0, 6),
    "cfg2": (2, 6, 6, 2, 100.0, 6),
    "cfg3": (3, 2, 4, 1, 100.0, 10),
    "cfg4": (3, 7, 3, 2, 1000.0, 10),
    "cfg4s": (3, 7, 2, 2, 1000.0, 10),   # config 4 truncated by one level (N = 28), for quick tuning runs
    "cfg4t": (3, 7, 1, 2, 1000.0, 10),   # two levels, N = 14 (173 k dofs): the per-rank share of config 4's level 2 at 8 GPUs
    # the coarse grid of the reference's largest defined run (ldc3d [P1+FB]^3, baseN 18, examples/generate_submission:10-23:
    # 243 k coarse dofs -- 473 GB as a dense inverse, hence the multifrontal coarse solver) with two of its four refinements
    "cfg6": (3, 18, 2, 1, 100.0, 10),
    "tiny": (3, 2, 1, 2, 1000.0, 4),
    # config 5 on ONE GPU: bfs3d channel, Scott-Vogelius [P3]^3 on Alfeld-split meshes, macro-star patches (up to 2175 dofs),
    # macro-cell transfer blocks of 390; baseN 1 (114 tets), nref 2 -> 441 k velocity dofs, 33 GB of patch inverses
    "cfg5": ("sv", 1, 2, 3, 500.0, 10),
    "cfg5s": ("sv", 1, 1, 3, 500.0, 10),
    "cfg5L": ("sv", 1, 3, 3, 500.0, 10),   # one more refinement: 3.4 M velocity dofs, ~40 GB of condensed factors on ONE GPU
    # config 5 on the reference's OWN channel mesh (examples/bfs3d/coarse60.msh, gmsh 2.2 ASCII: 299 nodes, 912 tets, physical
    # tags 1 / 2 / 3; kept as a workload input under data/meshes/): unstructured, macro stars up to 3615 dofs, a
    # 56 784-dof coarse grid (multifrontal solver).  One refinement: 440 022 velocity dofs; two: 3.47 M
    "cfg5m": ("sv", "bfs3d_coarse60.msh", 1, 3, 500.0, 10),
    "cfg5mL": ("sv", "bfs3d_coarse60.msh", 2, 3, 500.0, 10),
}
MESH_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "meshes")


def config_mesh(cfg):
    """The gmsh file of a config (None: structured); --mesh overrides it for the Scott-Vogelius configs."""
    base = CONFIGS[cfg][1]
    if BFS3D_MESH:
        return BFS3D_MESH
    return os.path.join(MESH_DIR, base) if isinstance(base, str) else None
SETUP_KEYS = ("comm_init", "partition", "localize", "upload_factor")     # DistMultigrid.setup_s, reported per rank
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)


def describe(cfg):
    dim, baseN, nref, ke, Re, k = CONFIGS[cfg]
    if dim == "sv":
        msh = config_mesh(cfg)
        base = "gmsh mesh %s" % os.path.basename(msh) if msh else "structured 10x2x1 channel with step, baseN %d" % baseN
        return ("bfs3d Scott-Vogelius [P%d]^3 on Alfeld-split meshes (%s, nref %d), "
                "Re=%g, gamma=1e4, FGMRES(%d)+macro-star patches" % (ke, base, nref, Re, k))
    el = "[P2]^2" if dim == 2 else ("[P1+FB]^3" if ke == 1 else "[P2+FB]^3")
    return "ldc%dd %s-P0, N=%d (baseN %d, nref %d), Re=%g, gamma=1e4, FGMRES(%d)+star patches" % (
        dim, el, baseN * 2 ** nref, baseN, nref, Re, k)


BFS3D_MESH = None      # --mesh: a gmsh 2.2 ASCII channel mesh for the Scott-Vogelius configs instead of the structured stand-in


def build_problem(cfg, verbose, lazy=False):
    """lazy: rank-local generation (alfi_amd.lazy) -- integers (meshes, numbering, graphs, patches) now, operator and
    transfer values only for the rows a rank's partition asks for; used by the multi-GPU leg."""
    from alfi_amd.problem import TwoDimLidDrivenCavityProblem, ThreeDimLidDrivenCavityProblem, build_hierarchy
    dim, baseN, nref, ke, Re, k = CONFIGS[cfg]
    if dim == "sv":
        from alfi_amd.problem import ThreeDimBackwardsFacingStepProblem
        from alfi_amd.sv import build_sv_hierarchy
        msh = config_mesh(cfg)
        lv, tr = build_sv_hierarchy(ThreeDimBackwardsFacingStepProblem(1 if msh else baseN, msh=msh), nref, ke, Re=Re)
        return lv, tr, k
    prob = TwoDimLidDrivenCavityProblem(baseN) if dim == 2 else ThreeDimLidDrivenCavityProblem(baseN)
    lv, tr = build_hierarchy(prob, nref, ke, Re=Re, verbose=verbose, lazy=lazy)
    return lv, tr, k


def apply_kernel_name(L):
    from alfi_amd.hip import condense_patches
    if condense_patches(L):
        # one apply = three launches (kernels_bigpatch.hip): a workgroup per patch on launches of >= 1024 patches, chunks of
        # groups (cond_g* kernels) on smaller ones
        if len(L.patch_ptr) - 1 < 1024:
            return "cond_gfront_kernel + cond_gsigma_kernel + cond_gback_kernel"
        return "cond_front_kernel + cond_sigma_kernel + cond_back_kernel"
    n = np.diff(L.patch_ptr).max()
    return "big_apply_kernel" if n > 160 else ("patch_apply_il_kernel" if n <= 32 else "patch_apply_kernel")


def vcycle_bytes(levels, dmg, k):
    """Algorithmic HBM bytes of one V-cycle and of one patch_apply_kernel launch per level."""
    total = 0.0
    per_level_apply = {}
    for L, dl in zip(levels, dmg.levels):
        if L.level == 0:
            # dense inverse: one n x n GEMV; multifrontal factors: both sweeps read them once, twice with the refinement step
            fb = float(dl.coarse_factor_bytes())
            total += fb if fb == 8.0 * L.n * L.n else 2.0 * fb
            continue
        npatch, sum_n, sum_n2 = dl.patch_stats()
        fbytes = float(dl.factor_bytes())                      # 8 sum n_p^2 for dense inverses, less for condensed factors
        b_apply_kernel = fbytes + 20.0 * sum_n                 # factors + int32 dofs + gathered x + staged result
        b_patch = fbytes + 28.0 * sum_n + 8.0 * L.n            # + dof-wise sum (read staged, write y)
        bs = L.bs
        b_spmv = (8.0 * bs * bs + 4.0) * L.A.nnzb + 4.0 * (L.A.nbrows + 1) + 16.0 * L.n
        b_blas1 = 8.0 * L.n * (k * k + 3 * k + 2)
        total += 2 * (k * (b_patch + b_spmv) + b_blas1) + b_spmv
        per_level_apply[dl.id] = b_apply_kernel
    return total, per_level_apply


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a version banner at
    communicator creation): point file descriptor 1 at stderr for the rest of the run and return a function that writes
    to the real stdout."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        os.write(real, (line + "\n").encode())
    return emit


class Stages(object):
    """Per-rank stage markers: one stderr line per stage and, when the launcher set ALFI_BENCH_STAGE_DIR, a file the
    launcher's watchdog reads.  A rank that hangs (a communicator that never forms, an exchange whose groups do not pair)
    is then reported with the stage it hangs in instead of a silent driver timeout."""

    def __init__(self, rank):
        self.rank, self.t0, self.cur, self.seen = rank, time.time(), "start", []
        d = os.environ.get("ALFI_BENCH_STAGE_DIR")
        self.path = os.path.join(d, "rank%d.stage" % rank) if d else None
        spec = os.environ.get("ALFI_BENCH_TEST_HANG", "")        # "<rank>:<stage>": fault injection for the watchdog test
        self.hang = tuple(spec.split(":", 1)) if ":" in spec else None

    def __call__(self, name):
        self.cur = name
        dt = time.time() - self.t0
        self.seen.append((name, round(dt, 1)))
        sys.stderr.write("[alfi bench] rank %d: stage %s (+%.1f s)\n" % (self.rank, name, dt))
        sys.stderr.flush()
        if self.path:
            with open(self.path, "w") as f:
                f.write("%s %.1f\n" % (name, dt))
        if self.hang and self.hang == (str(self.rank), name):
            while True:                                          # never reached in production
                time.sleep(3600)


def bench_timeout_s():
    return float(os.environ.get("ALFI_BENCH_TIMEOUT_S", "900"))


def start_rank_watchdog(stages):
    """In-rank half of the watchdog (covers the `torch.distributed.run` launch, where no parent of ours watches): a daemon
    thread that ends THIS process with the stage it is stuck in once ALFI_BENCH_TIMEOUT_S have passed.  The blocking calls
    of a rank (ncclCommInitRank, stream synchronisation, torch.distributed) run inside ctypes / torch with the GIL
    released, so the thread gets to run while the main thread hangs."""
    import threading
    limit = bench_timeout_s()
    if limit <= 0 or os.environ.get("ALFI_BENCH_RANK_WATCHDOG", "1") == "0":
        return None
    done = threading.Event()

    def watch():
        if done.wait(limit):
            return
        sys.stderr.write("[alfi bench] rank %d: WATCHDOG: no result after %.0f s, stuck in stage %r; stages so far %r\n"
                         % (stages.rank, limit, stages.cur, stages.seen))
        sys.stderr.flush()
        os._exit(124)
    th = threading.Thread(target=watch, name="alfi-bench-watchdog", daemon=True)
    th.start()
    return done


def main_distributed(args, rank, world, local_rank):
    """One process per GPU: the fixed config-4 mesh partitioned over the ranks (strong scaling), halos and reductions over
    RCCL issued by the library itself.  Every rank builds the integer side of the hierarchy (meshes, numbering, graphs,
    patches: what the partitioner needs) and assembles operator / transfer values for its own rows only."""
    stage = Stages(rank)
    watchdog_done = start_rank_watchdog(stage)
    stage("import_torch")
    import torch
    import torch.distributed as dist
    emit = claim_stdout()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs MI355X GPUs: the HIP path has no CPU fallback")
    # ALFI_DIST_BACKEND=gloo: functional check of this leg with several ranks sharing one GPU (halos staged through the
    # host) -- never a performance number
    backend = os.environ.get("ALFI_DIST_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    stage("process_group_init")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    from alfi_amd._hostlib import cpu_share
    os.environ.setdefault("ALFI_HOST_THREADS", str(max(1, cpu_share() // world)))    # host generator threads of this rank
    from alfi_amd.dist import DistMultigrid
    stage("host_generation")
    t0 = time.time()
    lazy = CONFIGS[args.config][0] != "sv" and os.environ.get("ALFI_DIST_GLOBAL_GENERATION") != "1"
    # the integer side is generated ONCE per node (rank 0, with all of the job's host threads) and mapped copy-on-write by
    # the other ranks (alfi_amd.shared); ALFI_DIST_SHARED_GENERATION=0: every rank generates its own copy
    shared = world > 1 and os.environ.get("ALFI_DIST_SHARED_GENERATION", "1") != "0"
    if shared:
        from alfi_amd import _hostlib
        from alfi_amd.shared import build_shared
        nthr = lambda n: _hostlib.lib().alfi_host_set_num_threads(int(n))
        def bcast(x):
            box = [x]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        lv, tr, k = build_shared(lambda: build_problem(args.config, args.verbose, lazy=lazy), rank, bcast, dist.barrier,
                                 args.config, set_threads=nthr,
                                 all_threads=cpu_share(), my_threads=max(1, cpu_share() // world))
    else:
        lv, tr, k = build_problem(args.config, args.verbose and rank == 0, lazy=lazy)
    t_gen = time.time() - t0

    t0 = time.time()
    dmg = DistMultigrid(lv, tr, k, robust_restriction=args.restriction,
                        min_dofs=int(os.environ.get("ALFI_DIST_MIN_DOFS", "400000")), verbose=args.verbose,
                        force_distributed=os.environ.get("ALFI_DIST_FORCE") == "1", on_stage=stage)
    dmg.sync()
    t_setup = time.time() - t0
    stage("first_cycle")
    L = lv[-1]
    b = np.random.default_rng(0).standard_normal(L.n)
    b[L.bc_dofs] = 0.0
    db, dx = dmg.local_vec(b), dmg.local_vec()
    ctx = dmg.ctx
    for _ in range(args.warmup):
        dmg.vcycle(db, dx)
    dmg.sync()
    stage("timed_cycles")
    # events around the dominant kernel only (the roofline block needs them): the small distributed levels are bound by
    # per-launch latency and every event record adds to it; the exchanges are timed in one extra cycle afterwards
    prof_mode = {"0": False, "1": True}.get(os.environ.get("ALFI_BENCH_PROF", "3"), 3)
    ctx.prof_enable(prof_mode)
    ctx.prof_reset()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dmg.vcycle(db, dx)
    dmg.sync()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ctx.prof_enable(False)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    stage("post_measurements")
    # convergence sanity on the timed iterate: global residual norm
    dr = dmg.local_vec()
    with torch.cuda.stream(dmg.stream):
        dmg.levels[-1].residual(db, dx, dr)
    r2 = torch.tensor([float(np.sum(dmg.owned(dr) ** 2))], dtype=torch.float64, device="cuda")
    dist.all_reduce(r2)
    res = float(np.sqrt(r2.item()) / np.linalg.norm(b))

    fin = dmg.levels[-1]
    npatch, sum_n, sum_n2 = fin.patch_stats()
    prof = ctx.prof_get()
    ms_f, cnt_f = ctx.prof_get(fin.id)["PATCH_APPLY"]
    comm_ms = prof["COMM"][0] / args.steps
    if True:                                         # device time of the exchange points: one extra, untimed cycle
        ctx.prof_enable(2)
        ctx.prof_reset()
        ctx.comm_stats(reset=True)
        dmg.vcycle(db, dx)
        dmg.sync()
        n_halo, n_red, sent = ctx.comm_stats()
        comm_ms = ctx.prof_get()["COMM"][0]
        ctx.prof_enable(False)
        prof = dict(prof, COMM=(comm_ms * args.steps, prof["COMM"][1]))
    # the Amdahl term of the partitioned cycle: device time this rank spends on levels ONE rank owns entirely (the coarse
    # grid and the levels below min_dofs live on rank 0, as the reference telescopes its coarse solve, solver.py:354-358),
    # every event class of one more untimed, fully instrumented cycle
    ctx.prof_enable(True)
    ctx.prof_reset()
    dmg.vcycle(db, dx)
    dmg.sync()
    single_ms = 0.0
    for dl, p in zip(dmg.levels, dmg.parts[dmg.lmin:]):
        if not p.distributed and p.nb_own > 0:
            single_ms += sum(v[0] for v in ctx.prof_get(dl.id).values())
    from alfi_amd._lib import EVENTS
    ev_v = [ctx.prof_get()[kname][0] for kname in EVENTS]          # this rank's device time per event class, one V-cycle
    ctx.prof_enable(False)
    # the cycle the reference actually applies is the FULL cycle (pc_mg_type full, solver.py:366): every level below the finest
    # is visited more often than in a V-cycle, so the single-owner levels weigh more.  Two timed F-cycles (max over ranks), then
    # one fully instrumented one for the per-rank event table and the single-owner share.
    dxf = dmg.local_vec()
    dmg.fcycle(db, dxf)
    dmg.sync()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        dmg.fcycle(db, dxf)
    dmg.sync()
    torch.cuda.synchronize()
    f_ms = 1e3 * (time.perf_counter() - t0) / 2
    ctx.prof_enable(True)
    ctx.prof_reset()
    dmg.fcycle(db, dxf)
    dmg.sync()
    single_f_ms = 0.0
    for dl, p in zip(dmg.levels, dmg.parts[dmg.lmin:]):
        if not p.distributed and p.nb_own > 0:
            single_f_ms += sum(v[0] for v in ctx.prof_get(dl.id).values())
    ev_f = [ctx.prof_get()[kname][0] for kname in EVENTS]
    ctx.prof_enable(False)
    lib_rank, lib_world = ctx.comm_size() if dmg.transport == "rccl" else (rank, None)
    nbr_max = max([int(np.count_nonzero((p.send_counts > 0) | (p.recv_counts > 0))) for p in dmg.parts] + [0])
    outer = None
    if args.outer:
        # one Newton-step linear solve on the partitioned levels (alfi_amd.dist.DistSaddle), outside the timed region
        from alfi_amd.dist import DistSaddle
        if CONFIGS[args.config][0] == "sv":      # discontinuous P2 pressure, block DGMassInv; bfs3d has an outflow: no nullspace
            from alfi_amd.sv import build_sv_pressure_coupling
            Bm, _, Minv = build_sv_pressure_coupling(L)
            sad = DistSaddle(dmg, Bm, None, L.V.cell_nodes, L.nu, L.gamma, remove_constant_nullspace=False, mass_inv=Minv)
        else:
            from alfi_amd.problem import build_pressure_coupling
            Bm, vol = build_pressure_coupling(L)
            sad = DistSaddle(dmg, Bm, vol, L.V.cell_nodes, L.nu, L.gamma, remove_constant_nullspace=True)
        rtol, atol = (1e-9, 1e-10) if L.bs == 2 else (1e-8, 1e-8)
        rhs = torch.tensor(np.concatenate([b[dmg.fine.part.own_dofs()], np.zeros(sad.np_own)]), dtype=torch.float64,
                           device=dmg.device)
        dmg.sync()
        dist.barrier()
        t0 = time.perf_counter()
        _, its, rn = sad.solve(rhs, rtol=rtol, atol=atol)
        dmg.sync()
        outer = {"what": "one Newton-step linear solve on the partitioned levels: host-driven FGMRES(30) + fieldsplit Schur "
                         "full, 2 PCMG full cycles per iteration", "iterations": its, "seconds": time.perf_counter() - t0,
                 "true_residual_norm": rn, "rhs_norm": float(np.linalg.norm(b)), "rtol": rtol, "atol": atol,
                 "pressure_dofs": int(Bm.shape[0])}
        sad.close()
    bytes_apply = float(fin.factor_bytes()) + 20.0 * sum_n
    # with the halo overlap one apply is three launches of patch_apply_kernel (interior half | boundary | interior half):
    # account per apply, not per launch
    applies = args.steps * 2 * k
    local_gbs = bytes_apply * applies / (ms_f * 1e-3) / 1e9 if ms_f > 0 else 0.0
    rss_gb = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
    stats = torch.tensor([float(npatch), float(dmg.n_own), float(dmg.n_loc - dmg.n_own), local_gbs, comm_ms, rss_gb,
                          t_gen, t_setup, float(n_halo), float(n_red), 8e-6 * sent, float(nbr_max), single_ms,
                          float(lib_world if lib_world is not None else -1)] + [dmg.setup_s.get(kk, 0.0) for kk in SETUP_KEYS]
                         + [f_ms, single_f_ms] + ev_v + ev_f,
                         dtype=torch.float64, device="cuda")
    allstats = [torch.zeros_like(stats) for _ in range(world)]
    dist.all_gather(allstats, stats)
    if rank == 0:
        vps = args.steps / elapsed
        per_rank = [[float(v) for v in t.tolist()] for t in allstats]
        o_f = 14 + len(SETUP_KEYS)                # [f_ms, single_f_ms, events of a V-cycle, events of an F-cycle]
        ne = len(EVENTS)
        out = {
            "metric": ("V-cycles/sec on bfs3d SV P3-P2dg (DoF*smooths/sec in dof_smooths_per_s)" if CONFIGS[args.config][0] == "sv"
                   else "V-cycles/sec on ldc%dd P2-P0 (DoF*smooths/sec in dof_smooths_per_s)" % CONFIGS[args.config][0]),
            "value": vps, "unit": "V-cycles/s", "n_gpus": world,
            # size of the communicator the LIBRARY exchanges over (alfi_ctx_comm_size), agreed by every rank; with the
            # callback test transport there is no such communicator and the process group's size is reported instead
            "n_ranks_seen": (int(per_rank[0][13]) if per_rank[0][13] >= 0 and len(set(r[13] for r in per_rank)) == 1
                             else int(dist.get_world_size()) if per_rank[0][13] < 0 else -1),
            "n_ranks_seen_source": "alfi_ctx_comm_size (library-owned RCCL communicator)" if per_rank[0][13] >= 0
                                   else "torch.distributed world size (callback test transport)",
            "process_group_size": int(dist.get_world_size()),
            "neighbours_max": int(max(r[11] for r in per_rank)),
            "single_owner_levels_ms": round(per_rank[0][12], 3),
            # the FULL cycle (what the reference applies per MG application, solver.py:366): wall time (max over ranks, two
            # cycles between barriers) and rank 0's device time on the levels it owns alone -- the Amdahl term of that cycle
            "fcycle_ms": round(max(r[o_f] for r in per_rank), 3),
            "fcycle_single_owner_levels_ms": round(per_rank[0][o_f + 1], 3),
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": describe(args.config), "name": args.config, "velocity_dofs": int(L.n),
                       "levels": len(lv), "patches_finest": int(len(L.patch_ptr) - 1),
                       "cycle": "V(k,k), 1 cycle per step", "robust_restriction": bool(args.restriction),
                       "parallelism": "mesh partition over %d GPUs (Morton boxes, RCCL halos + all-reduce)" % world,
                       "backend": backend, "transport": dmg.transport,
                       "generation": ("integers once per node, shared through /dev/shm; " if shared else "integers on every rank; ")
                                     + ("values rank-local (alfi_amd.lazy)" if lazy else "values global"),
                       "distributed_levels": [int(p.level) for p in dmg.parts if p.distributed]},
            "dof_smooths_per_s": L.n * 2 * k * vps,
            "roofline": {"kernel": apply_kernel_name(L), "bound": "hbm", "achieved": local_gbs, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": local_gbs / HBM_PEAK_GBS, "traffic": None,
                         "note": "rank 0's finest-level applies (its share of the patches; one apply = up to three "
                                 "launches around the halo exchanges), HIP events",
                         "avg_apply_us": 1e3 * ms_f / max(applies, 1), "applies": int(applies), "launches": int(cnt_f)},
            "events_ms_rank0": {kname: round(v[0], 3) for kname, v in prof.items()},
            "per_rank": {"patches_finest": [r[0] for r in per_rank], "owned_dofs": [r[1] for r in per_rank],
                         "ghost_dofs": [r[2] for r in per_rank], "patch_apply_GBps": [round(r[3], 1) for r in per_rank],
                         "comm_ms_per_cycle": [round(r[4], 3) for r in per_rank],
                         # exchange points of ONE V-cycle as the library counts them (alfi_ctx_comm_stats): halo exchanges
                         # (forward + reverse-add; each one grouped send/recv with the level's real neighbours), all-reduces
                         # (<= 12 doubles each), MB this rank sends
                         "halo_exchanges_per_cycle": [int(r[8]) for r in per_rank],
                         "allreduces_per_cycle": [int(r[9]) for r in per_rank],
                         "MB_sent_per_cycle": [round(r[10], 2) for r in per_rank],
                         "neighbours_max": [int(r[11]) for r in per_rank],
                         "single_owner_levels_ms": [round(r[12], 3) for r in per_rank],
                         "fcycle_ms": [round(r[o_f], 3) for r in per_rank],
                         "fcycle_single_owner_levels_ms": [round(r[o_f + 1], 3) for r in per_rank],
                         # device time per event class and rank (HIP events on each rank's stream), one fully instrumented cycle
                         "events_ms_vcycle": {kname: [round(r[o_f + 2 + i], 3) for r in per_rank] for i, kname in enumerate(EVENTS)},
                         "events_ms_fcycle": {kname: [round(r[o_f + 2 + ne + i], 3) for r in per_rank] for i, kname in enumerate(EVENTS)},
                         "setup_s": {kk: [round(r[14 + i], 2) for r in per_rank] for i, kk in enumerate(SETUP_KEYS)}},
            "rel_residual_after_timed_cycles": res,
            "setup_s": {"host_generation": round(max(r[6] for r in per_rank), 1),
                        "partition_and_device_setup": round(max(r[7] for r in per_rank), 1),
                        "host_peak_rss_GB_per_rank": [round(r[5], 1) for r in per_rank],
                        "note": "max over ranks; generation = meshes, numbering, graphs, patches (every rank); setup = "
                                "partition, assembly of the rank's own rows, upload, patch inversion"},
            "cpu_baseline": {"value": None, "unit": "V-cycles/s", "cores": 0, "kind": "port",
                             "sample": "reported at N=1 only"},
        }
        if outer is not None:
            out["outer_solve"] = outer
        assert out["neighbours_max"] <= min(world - 1, 7) or world > 8, "a rank exchanges with more than 7 neighbours"
        emit(json.dumps(out))
    stage("teardown")
    dist.barrier()
    dmg.close()
    dist.destroy_process_group()
    if watchdog_done is not None:
        watchdog_done.set()
    stage("done")


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N rank processes as children -- one per GPU, the same
    command line, the rank environment torch.distributed.run would export -- relay rank 0's JSON line and exit non-zero
    if any rank fails.  Nothing in this (parent) process touches a GPU: `device_count()` reads the topology without
    creating a context, and the children are new processes, not an exec of an initialised one."""
    import signal
    import socket
    import subprocess
    import torch
    ndev = torch.cuda.device_count()
    shared = os.environ.get("ALFI_DIST_BACKEND", "nccl") != "nccl"     # gloo: functional runs with ranks sharing a GPU
    if ndev < n and not shared:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (RCCL needs one GPU per rank; ALFI_DIST_BACKEND=gloo "
                         "runs the ranks on a shared GPU as a functional check)" % (n, ndev))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    import tempfile
    stage_dir = tempfile.mkdtemp(prefix="alfi_bench_stages_")
    limit = bench_timeout_s()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ALFI_BENCH_STAGE_DIR=stage_dir)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: the only mode the host driver supports
        # children are NEW processes started before anything in this parent touched a GPU -- never an exec of an
        # initialised one; they stay in this process's group and session, so whoever ends the launcher's group ends them too
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, cwd=os.getcwd()))

    def last_stages():
        out = []
        for r in range(n):
            try:
                out.append("rank %d: %s" % (r, open(os.path.join(stage_dir, "rank%d.stage" % r)).read().strip()))
            except OSError:
                out.append("rank %d: (no stage reported -- still importing / starting)" % r)
        return "; ".join(out)

    def end_children():
        for p in procs:                                            # exactly the children started above, by PID
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    def on_term(signum, frame):                                    # the launcher itself is being ended: take the ranks along
        sys.stderr.write("bench.py: signal %d; last stages -- %s\n" % (signum, last_stages()))
        end_children()
        raise SystemExit(128 + signum)
    signal.signal(signal.SIGTERM, on_term)

    line = None
    failed = None
    timed_out = False
    import selectors
    sel = selectors.DefaultSelector()
    sel.register(procs[0].stdout, selectors.EVENT_READ)
    buf = b""
    open_out = True
    t_start = time.time()
    while True:
        if open_out:
            for _ in sel.select(timeout=0.5):
                chunk = os.read(procs[0].stdout.fileno(), 65536)
                if not chunk:
                    open_out = False
                    sel.unregister(procs[0].stdout)
                buf += chunk
        else:
            time.sleep(0.2)
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = bad[0]
            break
        if all(c == 0 for c in codes) and not open_out:
            break
        # watchdog (ALFI_BENCH_TIMEOUT_S, 0 = off): a communicator that never forms or an exchange whose send/recv groups
        # do not pair would otherwise spin until the driver's own limit and leave nothing to diagnose; the grace period
        # lets the ranks' own watchdogs (same limit) report first
        if limit > 0 and time.time() - t_start > limit + float(os.environ.get("ALFI_BENCH_GRACE_S", "15")):
            timed_out = True
            break
    if failed is not None or timed_out:
        stages_txt = last_stages()
        end_children()
        if timed_out:
            sys.stderr.write("bench.py: WATCHDOG: no result after %.0f s (ALFI_BENCH_TIMEOUT_S); last stages -- %s\n"
                             % (limit, stages_txt))
        else:
            sys.stderr.write("bench.py: rank %d exited with code %d; last stages -- %s\n" % (failed + (stages_txt,)))
        import shutil
        shutil.rmtree(stage_dir, ignore_errors=True)
        raise SystemExit(1)
    import shutil
    shutil.rmtree(stage_dir, ignore_errors=True)
    for cand in buf.decode(errors="replace").splitlines():
        if cand.startswith("{"):
            line = cand
    if line is None:
        sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
        raise SystemExit(1)
    print(line, flush=True)
    return 0


def main():
    # dmabuf IPC is the only mode the host driver of the GPU boxes supports (RCCL / tensor sharing across the rank processes);
    # before anything initialises the GPU, also when the ranks come from torch.distributed.run with a bare environment
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default=os.environ.get("ALFI_BENCH_CONFIG", "cfg4"), choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--patch-composition", default="additive", choices=["additive", "multiplicative"],
                    help="multiplicative: symmetrised Gauss-Seidel patch sweeps ordered by the problem's "
                         "relaxation_direction (alfi/solver.py:306-335); the headline metric is quoted on additive")
    ap.add_argument("--restriction", action="store_true",
                    help="Schoeberl's robust restriction (alfi/driver.py:41 `--restriction`, transfer.py:261-275): off by "
                         "default as in the reference's CLI; its production lines switch it on (examples/Makefile:6-16). "
                         "The single-GPU line always times the other setting as well and reports it as "
                         "`other_restriction_setting`")
    ap.add_argument("--outer", action="store_true",
                    help="also time one outer linear solve (FGMRES + fieldsplit Schur, alfi/solver.py:386-422) and "
                         "report it as `outer_solve`; not part of the headline metric")
    ap.add_argument("--graph", action="store_true",
                    help="after the timed (eager, event-instrumented) cycles also time the same cycles replayed as a "
                         "hipGraph (alfi_ctx_set_graph) and report them as `graph_replay`; single GPU only")
    ap.add_argument("--mesh", default=None,
                    help="config 5 only: gmsh 2.2 ASCII mesh of the channel (the reference's examples/bfs3d/coarse*.msh, "
                         "bfs3d.py:37-39) as the base mesh instead of the structured stand-in")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()
    global BFS3D_MESH
    BFS3D_MESH = args.mesh

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process only starts the N rank processes (before anything here touches a
        # GPU) and relays rank 0's JSON line; under torch.distributed.run the rank environment is already there
        return spawn_ranks(args.gpus)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force_dist = os.environ.get("ALFI_DIST_FORCE") == "1"       # 1-rank RCCL group: fixed cost of the exchange points
    if force_dist and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or args.gpus > 1 or force_dist:
        if world != args.gpus:
            raise SystemExit("--gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (args.gpus, world))
        return main_distributed(args, rank, world, local_rank)

    emit = claim_stdout()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(0)
    from alfi_amd import hip
    from alfi_amd import _lib as _alfi_lib

    t0 = time.time()
    lv, tr, k = build_problem(args.config, args.verbose)
    t_gen = time.time() - t0
    ctx = hip.Context(0)
    t0 = time.time()
    dmg = hip.Multigrid(ctx, lv, tr, k, robust_restriction=args.restriction, verbose=args.verbose)
    ctx.sync()
    t_setup = time.time() - t0
    L = lv[-1]
    wavefronts = None
    if args.patch_composition == "multiplicative":
        from alfi_amd.relaxation import OrderedRelaxation, Options
        wavefronts = []
        for Lh, dl in zip(lv[1:], dmg.levels[1:]):
            orl = OrderedRelaxation()
            orl.name = "Star"
            orl.opts = Options("", {"pc_patch_construction_Star_sort_order": "0+:1-"})   # ldc relaxation_direction
            iterset = orl.iteration_order(Lh.V.mesh.coords[Lh.patch_seeds])
            wavefronts.append(dl.set_multiplicative(iterset, True))
    b = np.random.default_rng(0).standard_normal(L.n)
    b[L.bc_dofs] = 0.0
    db, dx = ctx.vec(b), ctx.vec(L.n)

    for _ in range(args.warmup):
        dmg.vcycle(db, dx)
    ctx.sync()
    # HIP events around the dominant kernel only in the timed cycles (the roofline block is computed from them); the table
    # of all event classes and the SpMV figure come from one extra, untimed cycle.  ALFI_BENCH_PROF=0: no events at all
    # (tuning runs), 1: every class in the timed cycles as well.
    prof_mode = {"0": False, "1": True}.get(os.environ.get("ALFI_BENCH_PROF", "3"), 3)
    ctx.prof_enable(prof_mode)
    ctx.prof_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dmg.vcycle(db, dx)
    ctx.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ctx.prof_enable(False)
    timed_apply_all = ctx.prof_get()["PATCH_APPLY"]
    timed_apply = {dl.id: ctx.prof_get(dl.id)["PATCH_APPLY"] for dl in dmg.levels[1:]}
    ctx.prof_enable(True)
    ctx.prof_reset()
    dxi = ctx.vec(L.n)
    dmg.vcycle(db, dxi)
    ctx.sync()
    ctx.prof_enable(False)
    prof_all = ctx.prof_get()
    spmv_one_cycle = ctx.prof_get(dmg.levels[-1].id)["MATMULT"]

    # convergence sanity: the timed cycles must have reduced the residual (no skipped work)
    dr = ctx.vec(L.n)
    dmg.levels[-1].residual(db, dx, dr)
    res = float(np.linalg.norm(dr.get()) / np.linalg.norm(b))
    # pc_mg_type full (solver.py:366): one application of the reference's velocity-block preconditioner is an F-cycle;
    # reported next to the headline V-cycle figure (outside its timed region)
    dxf = ctx.vec(L.n)
    dmg.fcycle(db, dxf)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(2):
        dmg.fcycle(db, dxf)
    ctx.sync()
    fcycle_ms = 1e3 * (time.perf_counter() - t0) / 2

    # the other restriction setting, same levels / factors / transfers (a second alfi_mg over the same handles), untimed
    # warm-up + the same number of steps without events; reported next to the headline, never as `value`
    other = dmg.variant(robust_restriction=not args.restriction)
    dxo = ctx.vec(L.n)
    # (the host work in between lets the clocks drop, and they take ~0.1 s of load to come back: warm up for at least 0.3 s
    # -- with config 2's 5 ms cycles five warm-up cycles are not enough and the loop reads 7-8 ms)
    n_warm = max(args.warmup, 1, int(0.3 / max(elapsed / args.steps, 1e-4)))
    for _ in range(n_warm):
        other.vcycle(db, dxo)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(max(args.steps, 2)):
        other.vcycle(db, dxo)
    ctx.sync()
    other_ms = 1e3 * (time.perf_counter() - t0) / max(args.steps, 2)
    dmg.levels[-1].residual(db, dxo, dr)
    other_res = float(np.linalg.norm(dr.get()) / np.linalg.norm(b))
    other.close_handle()
    # the timed setting again without any HIP event (what a production cycle costs; on launch-bound configs the event
    # records around the dominant kernel lengthen the cycle)
    dxn = ctx.vec(L.n)
    for _ in range(n_warm):
        dmg.vcycle(db, dxn)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(max(args.steps, 2)):
        dmg.vcycle(db, dxn)
    ctx.sync()
    noevents_ms = 1e3 * (time.perf_counter() - t0) / max(args.steps, 2)

    graph_replay = None
    if args.graph:
        ctx.set_graph(True)
        dxg = ctx.vec(L.n)
        for _ in range(3):                      # eager, capture, first replay
            dmg.vcycle(db, dxg)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            dmg.vcycle(db, dxg)
        ctx.sync()
        tg = (time.perf_counter() - t0) / args.steps
        ctx.set_graph(False)
        ctx.prof_enable(False)
        dxe = ctx.vec(L.n)
        dmg.vcycle(db, dxe)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            dmg.vcycle(db, dxe)
        ctx.sync()
        te = (time.perf_counter() - t0) / args.steps
        graph_replay = {"ms_per_step": 1e3 * tg, "v_cycles_per_s": 1.0 / tg, "eager_without_events_ms_per_step": 1e3 * te,
                        "note": "same V-cycles replayed as one hipGraph; `value` above is the eager run with HIP events"}
    ms_per_step = 1e3 * elapsed / args.steps
    vps = args.steps / elapsed
    nlev = len(lv)
    smooths_per_cycle_finest = 2 * k
    total_bytes, apply_bytes = vcycle_bytes(lv, dmg, k)
    # dominant kernel: patch_apply_kernel, all launches of the timed region
    t_apply_ms, n_apply = timed_apply_all
    bytes_apply_total = 0.0
    per_level = {}
    for dl in dmg.levels[1:]:
        ms, cnt = timed_apply[dl.id]
        bytes_apply_total += apply_bytes[dl.id] * cnt
        per_level[dl.id] = (ms, cnt)
    achieved = bytes_apply_total / (t_apply_ms * 1e-3) / 1e9 if t_apply_ms > 0 else 0.0
    fin = dmg.levels[-1]
    ms_f, cnt_f = per_level[fin.id]
    finest_gbs = apply_bytes[fin.id] * cnt_f / (ms_f * 1e-3) / 1e9 if ms_f > 0 else 0.0
    t_spmv_ms, n_spmv = spmv_one_cycle
    bsz = L.bs
    b_spmv = (8.0 * bsz * bsz + 4.0) * L.A.nnzb + 4.0 * (L.A.nbrows + 1) + 16.0 * L.n
    spmv_gbs = b_spmv * n_spmv / (t_spmv_ms * 1e-3) / 1e9 if t_spmv_ms > 0 else 0.0

    # HBM traffic of the dominant kernel comes from rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE in separate runs,
    # scripts/gpu_check.sh + scripts/pmc_summary.py): hardware counters cannot be read from inside this process, so the
    # figure is the committed summary of the latest such pass -- NOT a measurement of this run; traffic_source says so.
    traffic, traffic_source = None, None
    pmc_file = os.path.join(ROOT, "profiles", "pmc_patch_apply_%s.json" % args.config)
    if os.path.exists(pmc_file):
        pmc = json.load(open(pmc_file))
        traffic_source = {"measured_in_this_run": False, "file": os.path.relpath(pmc_file, ROOT),
                          "collected": pmc.get("collected"), "commit": pmc.get("commit"), "box": pmc.get("box"),
                          "tag": pmc.get("tag")}
        # quoted only while the kernel sources are byte for byte what the counters were collected on (the summary carries
        # their SHA-256: the GPU box has no .git to ask); otherwise traffic stays null and the reason is given
        import hashlib
        csrc = os.path.join(ROOT, "alfi_amd", "csrc")
        want = pmc.get("kernel_sources_sha256")
        changed = None if not want else sorted(
            f for f, h in want.items()
            if not os.path.exists(os.path.join(csrc, f)) or hashlib.sha256(open(os.path.join(csrc, f), "rb").read()).hexdigest() != h)
        if want:       # ... and no kernel source has appeared since
            changed += sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".h")) and f not in want)
        if want and not changed:
            traffic = pmc.get("hbm_bytes_per_launch")
            traffic_source["kernel_sources"] = "unchanged since the collection"
        else:
            traffic_source["refused"] = ("the summary carries no source hashes (collected before round 4)" if not want
                                         else "kernel sources changed since the collection: " + ", ".join(changed))

    out = {
        "metric": ("V-cycles/sec on bfs3d SV P3-P2dg (DoF*smooths/sec in dof_smooths_per_s)" if CONFIGS[args.config][0] == "sv"
                   else "V-cycles/sec on ldc%dd P2-P0 (DoF*smooths/sec in dof_smooths_per_s)" % CONFIGS[args.config][0]),
        "value": vps,
        "unit": "V-cycles/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": describe(args.config), "name": args.config, "velocity_dofs": int(L.n),
                   "levels": nlev, "patches_finest": int(dmg.levels[-1].patch_stats()[0]),
                   "cycle": "V(k,k), 1 cycle per step", "parallelism": "1 GPU",
                   "robust_restriction": bool(args.restriction),
                   "patch_composition": args.patch_composition, "wavefronts_per_sweep": wavefronts},
        "ms_per_step_without_events": noevents_ms,
        "other_restriction_setting": {"robust_restriction": not args.restriction, "ms_per_step": other_ms,
                                      "v_cycles_per_s": 1e3 / other_ms, "rel_residual_after_cycles": other_res,
                                      "cycles": max(args.steps, 2) + n_warm,
                                      "note": "same hierarchy, the reference's `--restriction` flag flipped "
                                              "(examples/Makefile:6-16 passes it, driver.py:41 defaults it off); no events"},
        "dof_smooths_per_s": L.n * smooths_per_cycle_finest * vps,
        "patch_factor_GB": [round(dl.factor_bytes() / 1e9, 3) for dl in dmg.levels[1:]],
        "coarse_solver": {"kind": "dense inverse" if dmg.levels[0].coarse_factor_bytes() == 8 * lv[0].n * lv[0].n
                          else "multifrontal L D U + 1 refinement step", "dofs": lv[0].n,
                          "factor_GB": round(dmg.levels[0].coarse_factor_bytes() / 1e9, 3),
                          "probe_residual": dmg.levels[0].coarse_residual()},
        # residual probe of the patch factors of every smoothed level (worst before the pivoted repair, flagged, repaired,
        # worst afterwards): a smoother degraded by ill-conditioned patch operators shows here
        "patch_probe": [dict(zip(("worst", "flagged", "repaired", "worst_after"), dl.patch_check())) for dl in dmg.levels[1:]],
        "patch_factor_bytes_per_dof_finest": dmg.levels[-1].factor_bytes() / float(L.n),
        "fcycle_ms": fcycle_ms,
        "vcycle_algorithmic_GB": total_bytes / 1e9,
        "vcycle_hbm_frac_of_peak": total_bytes / 1e9 / (elapsed / args.steps) / HBM_PEAK_GBS,
        "roofline": {"kernel": apply_kernel_name(L), "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "avg_launch_us": 1e3 * t_apply_ms / max(n_apply, 1), "launches": int(n_apply),
                     "bytes_per_launch_avg": bytes_apply_total / max(n_apply, 1),
                     "finest_level_GBps": finest_gbs, "finest_level_avg_launch_us": 1e3 * ms_f / max(cnt_f, 1)},
        "spmv_finest": {"achieved_GBps": spmv_gbs, "frac": spmv_gbs / HBM_PEAK_GBS,
                        "avg_launch_us": 1e3 * t_spmv_ms / max(n_spmv, 1)},
        "events_ms": {kname: round(v[0], 3) for kname, v in prof_all.items()},
        "events_petsc_names": dict(_alfi_lib.PETSC_EVENT_NAMES),
        "events_note": "device time per event class of ONE extra, fully instrumented V-cycle after the timed region; on "
                       "launch-bound configurations (cfg2, cfg3) the event records lengthen the intervals they bracket and "
                       "the classes sum to more than the cycle: use the kernel trace (profiles/r02_kernel_trace_*.txt) there",
        "rel_residual_after_timed_cycles": res,
        "setup_s": {"host_generation": round(t_gen, 1), "device_setup_incl_patch_inversion": round(t_setup, 1),
                    "host_peak_rss_GB": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 1)},
    }
    if graph_replay is not None:
        out["graph_replay"] = graph_replay
    if args.outer:
        t0 = time.time()
        if CONFIGS[args.config][0] == "sv":      # discontinuous P2 pressure, block DGMassInv; bfs3d has an outflow: no nullspace
            from alfi_amd.sv import build_sv_pressure_coupling
            Bm, _, Minv = build_sv_pressure_coupling(L)
            sad = hip.Saddle(dmg, Bm, None, L.nu, L.gamma, remove_constant_nullspace=False, mass_inv=Minv)
        else:
            from alfi_amd.problem import build_pressure_coupling
            Bm, vol = build_pressure_coupling(L)
            sad = hip.Saddle(dmg, Bm, vol, L.nu, L.gamma, remove_constant_nullspace=True)
        t_b = time.time() - t0
        rtol, atol = (1e-9, 1e-10) if L.bs == 2 else (1e-8, 1e-8)          # solver.py:484-499
        dbb, dxx = ctx.vec(np.concatenate([b, np.zeros(Bm.shape[0])])), ctx.vec(L.n + Bm.shape[0])
        ctx.sync()
        t0 = time.perf_counter()
        its, rn = sad.solve(dbb, dxx, rtol, atol, 500, 30)
        ctx.sync()
        t_solve = time.perf_counter() - t0
        out["outer_solve"] = {"what": "one Newton-step linear solve: FGMRES(30) + fieldsplit Schur full, 2 PCMG full "
                                      "cycles per iteration, DGMassInv Schur approximation",
                              "iterations": its, "seconds": t_solve, "true_residual_norm": rn,
                              "rhs_norm": float(np.linalg.norm(b)), "rtol": rtol, "atol": atol,
                              "pressure_dofs": int(Bm.shape[0]), "divergence_setup_s": round(t_b, 1)}
        sad.close()
    baseline_failed = False
    if not args.no_cpu_baseline:
        try:
            from bench_cpu import cpu_baseline
            out["cpu_baseline"] = cpu_baseline(args.config, lv, tr, k, vps, robust_restriction=args.restriction)
        except Exception as e:
            # the GPU line is still printed (it is measured and valid), but the failure is LOUD: a non-null `error` field,
            # the traceback on stderr and a non-zero exit code after the line is out
            import traceback
            traceback.print_exc()
            out["cpu_baseline"] = {"value": None, "unit": "V-cycles/s", "cores": 0, "kind": "port",
                                   "sample": "failed", "error": "%s: %s" % (type(e).__name__, e)}
            baseline_failed = True
    else:
        out["cpu_baseline"] = {"value": None, "unit": "V-cycles/s", "cores": 0, "kind": "port", "sample": "skipped"}
    emit(json.dumps(out))
    dmg.close()
    ctx.close()
    if baseline_failed:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
