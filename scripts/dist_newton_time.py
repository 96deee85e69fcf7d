#!/usr/bin/env python3
"""Wall time of Newton steps with ALL device work on partitioned levels (alfi_amd.dist.DistNavierStokesSolver): the operator
refresh of every rank's own rows on the device (alfi_level_set_assembly on partitioned levels), the outer Krylov loop inside
the library (alfi_saddle_solve on a partitioned finest level).  Several ranks share the box's one GPU through the shared-memory
stand-in for librccl (tests/mock_rccl), so the times are FUNCTIONAL (one GPU does the work of N), not a scaling measurement.

    python scripts/dist_newton_time.py cfg4 --ranks 4 --re 10 100

The parent never touches the GPU: it starts the rank processes and relays rank 0's report; host assemblies during the Newton
loops are counted (must be 0) and the Newton / Krylov counts printed next to the single-GPU solver's when --compare is given."""
import argparse
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rank_main(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import bench
    from alfi_amd import _hostlib
    from alfi_amd.dist import DistNavierStokesSolver
    from alfi_amd.problem import TwoDimLidDrivenCavityProblem, ThreeDimLidDrivenCavityProblem
    dim, baseN, nref, ke, Re, k = bench.CONFIGS[args.config]
    prob = TwoDimLidDrivenCavityProblem(baseN) if dim == 2 else ThreeDimLidDrivenCavityProblem(baseN)
    t0 = time.time()
    s = DistNavierStokesSolver(prob, nref, ke, min_dofs=args.min_dofs, stabilisation_type="supg" if args.supg is not None else None,
                               stabilisation_weight=args.supg)
    if rank == 0:
        print("%s on %d ranks (transport %s): %d velocity + %d pressure dofs, setup %.1f s, device assembly %s, levels on "
              "this rank %d.." % (args.config, world, s.dmg.transport, s.n_u, s.n_p, time.time() - t0, s.device_assembly,
                                   s.dmg.lmin), flush=True)
    calls = []
    real, real_supg = _hostlib.assemble_bsr, _hostlib.supg
    _hostlib.assemble_bsr = lambda *a, **kw: (calls.append(1), real(*a, **kw))[1]
    _hostlib.supg = lambda *a, **kw: (calls.append(1), real_supg(*a, **kw))[1]
    for re in args.re:
        for kk in s.timings:
            s.timings[kk] = 0 if kk == "newton_steps" else 0.0
        dist.barrier()
        t0 = time.time()
        _, info = s.solve(re)
        wall = time.time() - t0
        n = max(s.timings["newton_steps"], 1)
        if rank == 0:
            print("Re %g: %d Newton steps, %d Krylov its, converged %s, wall %.2f s = %.2f s per Newton step "
                  "(assemble %.3f, factor %.3f, residual %.3f, solve %.3f per step); host assemblies so far: %d"
                  % (re, info["nonlinear_iter"], info["linear_iter"], info["converged"], wall, wall / n,
                     s.timings["assemble_s"] / n, s.timings["factor_s"] / n, s.timings["residual_s"] / n,
                     s.timings["solve_s"] / n, len(calls)), flush=True)
    got = [None] * world
    dist.all_gather_object(got, len(calls))
    if rank == 0:
        print("host assemblies during the Newton loops, per rank:", got, flush=True)
    s.close()
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--ranks", type=int, default=4)
    ap.add_argument("--re", type=float, nargs="+", default=[10.0, 100.0])
    ap.add_argument("--min-dofs", type=int, default=400000)
    ap.add_argument("--supg", type=float, default=None, metavar="WEIGHT",
                    help="SUPG stabilisation with this weight (the reference's production runs: 0.05)")
    ap.add_argument("--compare", action="store_true", help="also run the single-GPU solver (counts side by side)")
    ap.add_argument("--rank-process", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.rank_process:
        return rank_main(args)
    from tests.mock_rccl.build import build
    lib = build()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    threads = max(1, (os.cpu_count() or 8) // args.ranks)
    for r in range(args.ranks):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(args.ranks), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS=str(threads), ALFI_HOST_THREADS=str(threads),
                   ALFI_DIST_TRANSPORT="rccl", ALFI_RCCL_LIB=lib)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), args.config, "--ranks", str(args.ranks),
                                       "--min-dofs", str(args.min_dofs), "--rank-process"] +
                                      (["--supg", str(args.supg)] if args.supg is not None else []) + ["--re"] + [str(x) for x in args.re],
                                      env=env, cwd=ROOT))
    rc = [p.wait() for p in procs]
    if any(rc):
        raise SystemExit("rank exit codes %s" % rc)
    if args.compare:
        subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "newton_step_time.py"), args.config] +
                              (["--supg", str(args.supg)] if args.supg is not None else []) + ["--re"] + [str(x) for x in args.re],
                              cwd=ROOT)


if __name__ == "__main__":
    main()
