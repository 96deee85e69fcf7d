"""cpu_baseline leg of bench.py: the C/OpenMP oracle port (oracle/alfi_oracle.c, kind "port") timed on the host cores.

Default: the FULL hierarchy of the benchmarked configuration -- one warm-up and at least three timed V-cycles, as many as
fit ~25 s -- on the cores the cgroup grants the job.  Only when the host memory left in the cgroup cannot hold the port's
copy of the patch inverses (config 4: 35 GB) does it fall back to the hierarchy truncated by one level, and then the
value is labelled "extrapolated" and carries no GPU/CPU ratio."""
import os
import time

import numpy as np


def _host_memory_available():
    """Bytes this process may still allocate: cgroup limit minus current usage, capped by MemAvailable."""
    avail = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
    except OSError:
        pass
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        if lim != "max":
            cur = int(open("/sys/fs/cgroup/memory.current").read())
            left = int(lim) - cur
            avail = left if avail is None else min(avail, left)
    except (OSError, ValueError):
        pass
    return avail


def _port_bytes(levels, transfers):
    """Host memory the port allocates on top of the generated hierarchy: dense patch inverses, NUMA-local operator copies,
    staging and the interior-block inverses."""
    total = 0.0
    for L in levels:
        total += L.A.vals.nbytes + L.A.colidx.nbytes
        if L.level > 0:
            npd = np.diff(L.patch_ptr).astype(np.float64)
            total += 8.0 * float((npd * npd).sum()) + 24.0 * float(npd.sum())
    for T in transfers:
        total += 8.0 * T.blk_dofs.shape[0] * T.blk_dofs.shape[1] ** 2
    return total


def cpu_baseline(cfg_name, lv, tr, k, gpu_vps, robust_restriction=False):
    from oracle.c_oracle import CMultigrid, lib, spmv
    cores = lib().oracle_num_threads()
    need, avail = _port_bytes(lv, tr), _host_memory_available()
    full = os.environ.get("ALFI_CPU_BASELINE", "full") != "truncated" and (avail is None or 1.2 * need < avail)
    if full or len(lv) < 3:
        slv, strr, full = lv, tr, True
        what = "the full hierarchy"
    else:
        slv, strr = lv[:-1], tr[:-1]
        what = "levels 0..%d of the same hierarchy (host memory: need %.0f GB, %.0f GB left)" % (
            len(slv) - 1, need / 1e9, (avail or 0) / 1e9)
    t0 = time.time()
    mg = CMultigrid(slv, strr, k, robust_restriction=robust_restriction)
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
        if n >= 3 and (time.time() - t0 > 25.0 or n >= 10):
            break
    per_cycle = (time.time() - t0) / n
    # convergence of the sample itself (the port's own BSR product: no SciPy copy of the 1e9-entry operator at full size)
    res = float(np.linalg.norm(spmv(L.A, x, b=b, alpha=1.0)) / np.linalg.norm(b))
    if full:
        value = 1.0 / per_cycle
        sample = ("%d V-cycle(s) on %s (%d dofs), %.2f s each, after one warm-up cycle; patch inversion (%.1f s) excluded as "
                  "on the GPU side" % (n, what, L.n, per_cycle, t_setup))
        return {"value": value, "unit": "V-cycles/s", "cores": int(cores), "kind": "port", "sample": sample,
                "extrapolated": False, "gpu_over_cpu": gpu_vps / value if value > 0 else None,
                "rel_residual_after_sample": res, "cycles_in_sample": n + 1, "robust_restriction": bool(robust_restriction)}
    scale = L.n / lv[-1].n
    value = scale / per_cycle
    return {"value": value, "unit": "V-cycles/s", "cores": int(cores), "kind": "port", "extrapolated": True,
            "sample": "EXTRAPOLATED: %d V-cycle(s) on %s (%d of %d dofs), %.2f s each, scaled by the dof ratio %.4f"
                      % (n, what, L.n, lv[-1].n, per_cycle, scale),
            "gpu_over_cpu": None, "rel_residual_after_sample": res, "cycles_in_sample": n + 1,
            "robust_restriction": bool(robust_restriction)}
