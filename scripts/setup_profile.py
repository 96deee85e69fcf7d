#!/usr/bin/env python3
"""Where the set-up time of the Newton solver goes at a bench configuration's size (cProfile, cumulative):
  python scripts/setup_profile.py cfg4 [--supg 0.05]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--supg", type=float, default=None)
    ap.add_argument("--lines", type=int, default=45)
    args = ap.parse_args()
    import bench
    from alfi_amd.nssolver import HipNavierStokesSolver
    from alfi_amd.problem import TwoDimLidDrivenCavityProblem, ThreeDimLidDrivenCavityProblem
    dim, baseN, nref, ke, Re, k = bench.CONFIGS[args.config]
    prob = TwoDimLidDrivenCavityProblem(baseN) if dim == 2 else ThreeDimLidDrivenCavityProblem(baseN)
    pr = cProfile.Profile()
    t0 = time.time()
    pr.enable()
    s = HipNavierStokesSolver(prob, nref, ke, stabilisation_type="supg" if args.supg is not None else None,
                              stabilisation_weight=args.supg)
    s.ctx.sync()
    pr.disable()
    print("%s: solver set-up %.1f s (%d host threads)" % (args.config, time.time() - t0, len(os.sched_getaffinity(0))))
    pstats.Stats(pr).sort_stats("cumulative").print_stats(args.lines)
    s.close()


if __name__ == "__main__":
    main()
