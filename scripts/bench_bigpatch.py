#!/usr/bin/env python3
"""Throughput of the large-patch path (macro-star sized patches, n_p = 1275 as for Scott-Vogelius P3):
blocked Gauss-Jordan setup on the FP64 matrix cores and the one-workgroup-per-patch apply, on the config-4 operator of
level N = 28 with synthetic patches of 425 neighbouring nodes each (consecutive Morton indices).

  python scripts/bench_bigpatch.py [npatch] [nodes_per_patch]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from alfi_amd import hip

npatch = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
nodes_pp = int(sys.argv[2]) if len(sys.argv) > 2 else 425
lv, tr, k = bench.build_problem("cfg4s", False)
L = lv[-1]
bs = L.bs
free = np.flatnonzero(~L.V.bc_node_mask)
rng = np.random.default_rng(0)
starts = np.sort(rng.choice(len(free) - nodes_pp, npatch, replace=False))
dofs = np.concatenate([(free[s:s + nodes_pp][:, None] * bs + np.arange(bs)).ravel() for s in starts]).astype(np.int32)
ptr = (np.arange(npatch + 1, dtype=np.int64) * nodes_pp * bs)
n_p = nodes_pp * bs
ctx = hip.Context(0)
lvl = hip.Level(ctx, L.A, L.bc_dofs)
lvl.set_patches(ptr, dofs)
ctx.sync()
t0 = time.perf_counter()
lvl.factor()
ctx.sync()
t_f = time.perf_counter() - t0
t0 = time.perf_counter()
lvl.factor()
ctx.sync()
t_f = min(t_f, time.perf_counter() - t0)
flops = 2.0 * n_p ** 3 * npatch
x = rng.standard_normal(L.n)
dx, dy = ctx.vec(x), ctx.vec(L.n)
lvl.patch_apply(dx, dy)
ctx.sync()
ctx.prof_enable(True)
ctx.prof_reset()
for _ in range(10):
    lvl.patch_apply(dx, dy)
ctx.sync()
ms, cnt = ctx.prof_get()["PATCH_APPLY"]
bytes_apply = (8.0 * n_p * n_p + 20.0 * n_p) * npatch
# spot check of one inverse
p = npatch // 2
d = dofs[ptr[p]:ptr[p + 1]]
S = L.A.select_rows(np.unique(d // bs)).to_scipy().tocsc()[:, d].toarray()
Ap = S[(np.searchsorted(np.unique(d // bs), d // bs) * bs + d % bs)]
G = lvl.patch_inverse(p, n_p)
err = np.abs(G @ Ap - np.eye(n_p)).max()
Nref = np.linalg.inv(Ap)
err_lapack = np.abs(Nref @ Ap - np.eye(n_p)).max()
rel_to_lapack = np.abs(G - Nref).max() / np.abs(Nref).max()
cond = np.linalg.cond(Ap)
print({"npatch": npatch, "n_p": n_p, "inverse_GB": 8e-9 * n_p * n_p * npatch, "factor_s": t_f,
       "factor_TFLOPs": flops / t_f / 1e12, "apply_ms": ms / cnt, "apply_GBps": bytes_apply / (ms / cnt * 1e-3) / 1e9,
       "inverse_check_max_abs(Ainv A - I)": err, "same_for_LAPACK": err_lapack,
       "rel_diff_to_LAPACK_inverse": rel_to_lapack, "cond": cond})
