#!/usr/bin/env python3
"""The reference's acceptance experiment for the whole solver (examples/iters.py:40-78, examples/Makefile:5-16): average
outer Krylov iterations per Newton step for nref x Re with Reynolds continuation, on the synthetic lid-driven cavity, every
linear solve on the GPU (alfi_amd/nssolver.py).  Prints the nref / dofs / Re table of iters.py plus solve times.

  python scripts/iters_table.py --dim 2 --baseN 16 --nref-start 1 --nref-end 3 --re-max 1000
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alfi_amd.problem import TwoDimLidDrivenCavityProblem, ThreeDimLidDrivenCavityProblem
from alfi_amd.nssolver import HipNavierStokesSolver, run_solver


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=2)
    ap.add_argument("--baseN", type=int, default=16)
    ap.add_argument("--k", type=int, default=2)
    ap.add_argument("--nref-start", type=int, default=1)
    ap.add_argument("--nref-end", type=int, default=3)
    ap.add_argument("--re-max", type=int, default=1000)
    ap.add_argument("--gamma", type=float, default=1e4)
    ap.add_argument("--stabilisation-type", default="none", choices=["none", "supg"])
    ap.add_argument("--stabilisation-weight", type=float, default=None)
    args = ap.parse_args()
    # continuation as in alfi.driver.get_default_parser / run_solver: 0 (Stokes), 1, 10, 100, then steps of 250
    res = [0, 1, 10, 100] + list(range(250, args.re_max + 1, 250))
    res = [r for r in res if r <= args.re_max]
    tableres = [r for r in (10, 100, 1000, 5000, 10000) if r <= max(res)]
    rows = []
    for nref in range(args.nref_start, args.nref_end + 1):
        prob = TwoDimLidDrivenCavityProblem(args.baseN) if args.dim == 2 else ThreeDimLidDrivenCavityProblem(args.baseN)
        s = HipNavierStokesSolver(prob, nref, args.k, gamma=args.gamma, stabilisation_type=args.stabilisation_type,
                                  stabilisation_weight=args.stabilisation_weight)
        t0 = time.time()
        results = run_solver(s, res)
        rows.append((nref, s.n_u + s.n_p, results, time.time() - t0))
        s.close()
    print("nref\tdofs\t\t" + "\t".join("Re=%d" % r for r in tableres) + "\t(average Krylov iterations per Newton step)")
    for nref, dofs, results, t in rows:
        print("%d\t%.2e\t" % (nref, dofs) + "\t".join(
            "%.2f" % (results[r]["linear_iter"] / max(1, results[r]["nonlinear_iter"])) for r in tableres))
    print("nref\tdofs\t\t" + "\t".join("Re=%d" % r for r in tableres) + "\t(Newton steps; all converged: %s)"
          % all(results[r]["converged"] for _, _, results, _ in rows for r in res))
    for nref, dofs, results, t in rows:
        print("%d\t%.2e\t" % (nref, dofs) + "\t".join("%d" % results[r]["nonlinear_iter"] for r in tableres)
              + "\t[%.1f s for the whole continuation incl. host assembly]" % t)


if __name__ == "__main__":
    main()
