"""Algorithmic bytes of the condensed patch factors of a configuration's smoothed levels, by part: X_g (group inverses), B_g, W_g
and inv(Sigma), in the storage the kernels read (even leading dimensions, common.h: cond_group_doubles).  Host only.
usage: python scripts/cond_bytes.py cfg5"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import bench

lv, tr, k = bench.build_problem(sys.argv[1], False)
ld = lambda r: (r + 1) & ~1
for li in range(len(lv) - 1, 0, -1):
    L = lv[li]
    if getattr(L, "patch_groups", None) is None:
        continue
    # structural coupling of the block sparsity (node level), expanded to dofs
    Ab = sp.csr_matrix((np.ones(L.A.nnzb), np.asarray(L.A.colidx) & 0x7fffffff, np.asarray(L.A.rowptr)), shape=(L.A.nbrows, L.A.nbcols))
    Ab = (Ab + Ab.T).tocsr()
    bs = L.A.bs
    ptr, dofs, lab = np.asarray(L.patch_ptr), np.asarray(L.patch_dofs), np.asarray(L.patch_groups)
    X = B = W = S = 0
    for p in range(len(ptr) - 1):
        d, g = dofs[ptr[p]:ptr[p + 1]], lab[ptr[p]:ptr[p + 1]]
        sk = d[g < 0]
        s = len(sk)
        S += ld(s) * s
        if s == 0:
            continue
        for gg in np.unique(g[g >= 0]):
            rows = d[g == gg]
            m = len(rows)
            hit = np.asarray(Ab[np.unique(rows // bs)][:, np.unique(sk // bs)].sum(axis=0)).ravel() != 0
            sc = int(np.isin(sk // bs, np.unique(sk // bs)[hit]).sum())
            X += ld(m) * m
            B += ld(sc) * m
            W += ld(m) * sc
    print("%s level %d: %d patches; bytes X %.4e  B %.4e  W %.4e  inv(Sigma) %.4e  total %.4e; front reads X + B = %.4e, back W = %.4e"
          % (sys.argv[1], li, len(ptr) - 1, 8 * X, 8 * B, 8 * W, 8 * S, 8 * (X + B + W + S), 8 * (X + B), 8 * W))
