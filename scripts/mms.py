#!/usr/bin/env python3
"""The reference's convergence study (examples/mms.py:25-101): manufactured lid-driven-cavity solution, Reynolds continuation,
errors of velocity / velocity gradient / pressure and |div u_h| per refinement, observed convergence orders.  Every linear
solve on the GPU.

  python scripts/mms.py --dim 2 --baseN 8 --nref 4 --k 2 --discretisation pkp0 --re 1 10 100
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alfi_amd.mms import (TwoDimLidDrivenCavityMMSProblem, ThreeDimLidDrivenCavityMMSProblem, convergence_orders,  # noqa: E402
                          errors)
from alfi_amd.nssolver import HipNavierStokesSolver, run_solver                                                       # noqa: E402


def pressure_evaluator(s):
    """p_h at quadrature points for the solver's pressure space: P0 (one value per cell) or the discontinuous P_{k-1}."""
    L = s.levels[-1]
    if not s.sv:
        return {"p_cell": s.p}
    from alfi_amd.elements import NodalElement
    pel = NodalElement(L.V.dim, L.V.element.degree - 1, False)
    pc = s.p.reshape(L.V.mesh.num_cells, pel.nloc)
    return {"p_eval": lambda lam, cells: np.einsum("qj,cj->cq", pel.tabulate(lam)[0], pc[cells])}


def study(dim, baseN, nrefs, k, disc, res, gamma=1e4, verbose=True):
    out = {re: {n: [] for n in ("velocity", "velocitygrad", "pressure", "divergence")} for re in res}
    hs = []
    for nref in nrefs:
        prob = TwoDimLidDrivenCavityMMSProblem(baseN) if dim == 2 else ThreeDimLidDrivenCavityMMSProblem(baseN)
        s = HipNavierStokesSolver(prob, nref, k, gamma=gamma, discretisation=disc)
        hs.append(2.0 / (baseN * 2 ** nref))
        for re in res:
            z, info = s.solve(re)
            assert info["converged"], (nref, re, info)
            e = errors(s.levels[-1].V, s.u, prob, re, **pressure_evaluator(s))
            for n in out[re]:
                out[re][n].append(e[n])
            if verbose:
                print("nref %d Re %g: %s  (Newton %d, Krylov %d)" % (nref, re, {n: "%.3e" % v for n, v in e.items()},
                                                                    info["nonlinear_iter"], info["linear_iter"]), flush=True)
        s.close()
    return hs, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=2)
    ap.add_argument("--baseN", type=int, default=8)
    ap.add_argument("--nref", type=int, default=3)
    ap.add_argument("--k", type=int, default=2)
    ap.add_argument("--discretisation", default="pkp0", choices=["pkp0", "sv"])
    ap.add_argument("--gamma", type=float, default=1e4)
    ap.add_argument("--re", type=float, nargs="+", default=[1, 10, 100])
    a = ap.parse_args()
    hs, out = study(a.dim, a.baseN, range(1, a.nref + 1), a.k, a.discretisation, a.re, a.gamma)
    print("h =", hs)
    for re in a.re:
        print("Results for Re =", re)
        for n in ("velocity", "velocitygrad", "pressure", "divergence"):
            print("  |%s error|" % n, ["%.3e" % v for v in out[re][n]])
            if n != "divergence":
                print("  convergence orders:", np.round(convergence_orders(out[re][n]), 2))


if __name__ == "__main__":
    main()
