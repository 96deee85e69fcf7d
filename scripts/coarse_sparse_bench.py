"""Sparse (multifrontal) against dense coarse factorisation on ldc3d [P2+FB]^3 coarse grids of growing size.
usage: python scripts/coarse_sparse_bench.py N [N ...]   (env LEAF: nodes per leaf subdomain)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alfi_amd import hip                                                            # noqa: E402
from alfi_amd.problem import ThreeDimLidDrivenCavityProblem, build_hierarchy        # noqa: E402


def timed_solves(ctx, dl, n, reps=20):
    b = np.random.default_rng(0).standard_normal(n)
    bx, xx = ctx.vec(b), ctx.vec(n)
    dl.coarse_solve(bx, xx)
    ctx.sync()
    t0 = time.time()
    for _ in range(reps):
        dl.coarse_solve(bx, xx)
    ctx.sync()
    return (time.time() - t0) / reps, b, xx.get()


for N in [int(a) for a in sys.argv[1:]]:
    t0 = time.time()
    lv, _ = build_hierarchy(ThreeDimLidDrivenCavityProblem(N), 0, 2, Re=1000.0, patches=False)
    L = lv[0]
    tg = time.time() - t0
    ctx = hip.Context(0)
    dl = hip.Level(ctx, L.A, L.bc_dofs)
    t0 = time.time()
    res = dl.coarse_factor_sparse(L.V.node_coords, leaf_nodes=int(os.environ.get("LEAF", 0)))
    tf = time.time() - t0
    ts, b, x = timed_solves(ctx, dl, L.n)
    r = np.abs(L.A.to_scipy() @ x - b).max()
    line = {"N": N, "dofs": L.n, "gen_s": round(tg, 1), "sparse_factor_s": round(tf, 2),
            "sparse_GB": round(dl.coarse_factor_bytes() / 1e9, 3), "probe": res, "sparse_solve_ms": round(ts * 1e3, 3),
            "residual_inf": r, "dense_GB": round(8e-9 * L.n * L.n, 1)}
    if L.n <= 60000:
        t0 = time.time()
        resd = dl.coarse_factor()
        line["dense_factor_s"] = round(time.time() - t0, 2)
        td, _, xd = timed_solves(ctx, dl, L.n)
        line["dense_solve_ms"] = round(td * 1e3, 3)
        line["dense_probe"] = resd
        line["dense_residual_inf"] = np.abs(L.A.to_scipy() @ xd - b).max()
        line["sparse_vs_dense"] = np.abs(x - xd).max() / np.abs(xd).max()
    print(line, flush=True)
    dl.close()
    ctx.close()
