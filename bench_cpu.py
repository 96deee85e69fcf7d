"""cpu_baseline leg of bench.py: the C/OpenMP oracle port (oracle/alfi_oracle.c, kind "port") timed on the host cores.

Bounded sample: the same hierarchy truncated by one level (e.g. config 4: N = 28 instead of 56; every kernel of the
cycle is O(dofs), so V-cycles/s scales with the dof ratio), at least one and at most as many V-cycles as fit ~20 s."""
import os
import time

import numpy as np


def cpu_baseline(cfg_name, lv, tr, k, gpu_vps):
    from oracle.c_oracle import CMultigrid, lib
    cores = lib().oracle_num_threads()
    if len(lv) >= 3:
        slv, strr = lv[:-1], tr[:-1]
        what = "levels 0..%d of the same hierarchy" % (len(slv) - 1)
    else:
        slv, strr = lv, tr
        what = "the full hierarchy"
    t0 = time.time()
    mg = CMultigrid(slv, strr, k)
    t_setup = time.time() - t0
    L = slv[-1]
    b = np.random.default_rng(0).standard_normal(L.n)
    b[L.bc_dofs] = 0.0
    x = np.zeros(L.n)
    x = mg.vcycle(len(slv) - 1, b, x)          # warm-up
    n, t0 = 0, time.time()
    while True:
        x = mg.vcycle(len(slv) - 1, b, x)
        n += 1
        if time.time() - t0 > 15.0 or n >= 10:
            break
    per_cycle = (time.time() - t0) / n
    scale = L.n / lv[-1].n
    value = scale / per_cycle
    return {"value": value, "unit": "V-cycles/s", "cores": int(cores), "kind": "port",
            "sample": "%d V-cycle(s) on %s (%d of %d dofs), %.2f s each, scaled by the dof ratio %.4f; patch inversion "
                      "(%.1f s) excluded as on the GPU side" % (n, what, L.n, lv[-1].n, per_cycle, scale, t_setup),
            "gpu_over_cpu": gpu_vps / value if value > 0 else None}
