#!/usr/bin/env python3
"""Wall time of Newton steps with the operators refreshed ON THE DEVICE (alfi_level_assemble) against the host rediscretisation
+ re-upload, at a bench configuration's size:  python scripts/newton_step_time.py cfg4 [--host] [--re 10 100]

Prints per Reynolds number the Newton / Krylov counts and the split of the wall time into assembly, patch + coarse
factorisation, residual evaluation and linear solve (HipNavierStokesSolver.timings)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--host", action="store_true", help="host rediscretisation (the round-2 path)")
    ap.add_argument("--re", type=float, nargs="+", default=[10.0, 100.0])
    ap.add_argument("--supg", type=float, default=None, metavar="WEIGHT",
                    help="SUPG stabilisation with this weight (the reference's production runs: 0.05, generate_submission:18-20)")
    args = ap.parse_args()
    import bench
    from alfi_amd.nssolver import HipNavierStokesSolver
    from alfi_amd.problem import TwoDimLidDrivenCavityProblem, ThreeDimLidDrivenCavityProblem
    dim, baseN, nref, ke, Re, k = bench.CONFIGS[args.config]
    prob = TwoDimLidDrivenCavityProblem(baseN) if dim == 2 else ThreeDimLidDrivenCavityProblem(baseN)
    t0 = time.time()
    s = HipNavierStokesSolver(prob, nref, ke, device_assembly=not args.host,
                              stabilisation_type="supg" if args.supg is not None else None, stabilisation_weight=args.supg)
    print("%s: %d velocity + %d pressure dofs, setup %.1f s, device assembly %s" % (args.config, s.n_u, s.n_p, time.time() - t0,
                                                                                  s.device_assembly), flush=True)
    for re in args.re:
        for kk in s.timings:
            s.timings[kk] = 0 if kk == "newton_steps" else 0.0
        t0 = time.time()
        _, info = s.solve(re)
        wall = time.time() - t0
        n = max(s.timings["newton_steps"], 1)
        print("Re %g: %d Newton steps, %d Krylov its, converged %s, wall %.2f s = %.2f s per Newton step "
              "(assemble %.3f, factor %.3f, residual %.3f, solve %.3f per step)"
              % (re, info["nonlinear_iter"], info["linear_iter"], info["converged"], wall, wall / n,
                 s.timings["assemble_s"] / n, s.timings["factor_s"] / n, s.timings["residual_s"] / n, s.timings["solve_s"] / n),
              flush=True)
    s.close()


if __name__ == "__main__":
    main()
