#!/usr/bin/env python3
"""The two constants of the overlap on/off rule (alfi_amd.dist.overlap_rule), measured on ONE GPU with REAL RCCL:

(A) EXCHANGE_LATENCY_US -- a 1-rank communicator whose rank is its own neighbour (test hook alfi_ctx_comm_allow_self): device time
    of one halo exchange -- pack kernel, ONE group of ncclSend / ncclRecv on the library's stream, unpack kernel -- against
    the halo size.  The size-independent part is what an exchange costs before the first byte moves; the slope here is a
    device-local copy, NOT an xGMI link (the rule takes the link bandwidth from MI355X_MICROARCH.md).
(B) OVERLAP_FIXED_US -- a 1-rank RCCL process group with every exchange point forced on (empty halos): V-cycles with the
    overlapped smoother iteration (interior | boundary launches around asynchronous begin / end pairs, three exchanges) against
    the plain one (two exchanges), per smoother iteration of the overlapped levels.

  python scripts/overlap_rule.py [--config cfg3]      -> profiles/r05_overlap_rule.txt
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def exchange_latency():
    from alfi_amd import hip, _lib
    from alfi_amd.problem import BSR
    rows = []
    for n_ghost in (64, 1024, 16384, 131072, 1048576):
        bs, n_own = 3, max(4 * n_ghost, 4096)
        nb = n_own + n_ghost
        A = BSR(nb, nb, bs, np.arange(nb + 1, dtype=np.int32), np.arange(nb, dtype=np.int32), np.tile(np.eye(bs), (nb, 1, 1)))
        ctx = hip.Context(0)
        ctx.comm_init(_lib.comm_unique_id(), 0, 1)
        ctx.comm_allow_self(True)
        L = hip.Level(ctx, A, np.zeros(0, dtype=np.int32))
        L.set_partition(n_own, True, (np.arange(n_ghost) * 3 % n_own).astype(np.int32), None, None, n_ghost)
        L.set_neighbours([0], [n_ghost], [n_ghost])
        dv = ctx.vec(np.ones(nb * bs))
        for _ in range(10):
            L.halo_forward(dv)
        ctx.sync()
        ctx.prof_enable(True)
        ctx.prof_reset()
        reps = 50
        t0 = time.perf_counter()
        for _ in range(reps):
            L.halo_forward(dv)
        ctx.sync()
        wall = (time.perf_counter() - t0) / reps
        dev = ctx.prof_get()["COMM"][0] / reps
        ctx.prof_enable(False)
        rows.append((n_ghost * bs * 8, 1e3 * dev, 1e6 * wall))
        ctx.close()
    return rows


def overlap_fixed(config):
    import torch
    import torch.distributed as dist
    import bench
    from alfi_amd.dist import DistMultigrid
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    lv, tr, k = bench.build_problem(config, False)
    b = np.random.default_rng(0).standard_normal(lv[-1].n)
    b[lv[-1].bc_dofs] = 0.0
    out = {}
    for name, kw in (("plain", dict(overlap=False)), ("overlapped", dict(overlap=True, overlap_min_dofs=0))):
        dmg = DistMultigrid(lv, tr, k, min_dofs=1, force_distributed=True, transport="rccl", **kw)
        db, dx = dmg.local_vec(b), dmg.local_vec()
        for _ in range(3):
            dmg.vcycle(db, dx)
        dmg.sync()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            dmg.vcycle(db, dx)
        dmg.sync()
        out[name] = 1e3 * (time.perf_counter() - t0) / n
        nlev = sum(1 for p in dmg.parts[1:] if p.distributed)
        dmg.close()
    dist.destroy_process_group()
    iters = 2 * k * nlev                    # smoother iterations of the overlapped levels per V-cycle
    return out, iters, lv[-1].n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg3")
    args = ap.parse_args()
    print("(A) one halo exchange through a REAL 1-rank RCCL communicator, the rank its own neighbour (pack + group of ncclSend / "
          "ncclRecv + unpack; HIP events on the library's stream, 50 repetitions)")
    print("    %12s %14s %14s" % ("bytes", "device us", "host wall us"))
    rows = exchange_latency()
    for by, dev, wall in rows:
        print("    %12d %14.1f %14.1f" % (by, dev, wall))
    print("    -> EXCHANGE_LATENCY_US = %.1f (the smallest message)" % rows[0][1])
    res, iters, n = overlap_fixed(args.config)
    print("(B) %s (%d dofs), 1-rank RCCL group, every exchange point on (empty halos): V-cycle plain %.3f ms, overlapped %.3f ms; "
          "%d smoother iterations on the overlapped levels" % (args.config, n, res["plain"], res["overlapped"], iters))
    print("    -> OVERLAP_FIXED_US = %.1f per smoother iteration" % (1e3 * (res["overlapped"] - res["plain"]) / iters))
    from alfi_amd.dist import overlap_rule, EXCHANGE_LATENCY_US, OVERLAP_FIXED_US, XGMI_LINK_GBPS
    print("rule in alfi_amd/dist.py: overlap iff 2 (EXCHANGE_LATENCY_US + bytes / link) > OVERLAP_FIXED_US with %.1f, %.1f us, %.0f GB/s"
          % (EXCHANGE_LATENCY_US, OVERLAP_FIXED_US, XGMI_LINK_GBPS))
    for by, inner in ((50e3, 0.05e9), (450e3, 4.4e9), (1e6, 8.7e9), (2e6, 17e9), (8e6, 35e9)):
        print("    largest message %8.0f KB, interior patch inverses %5.2f GB -> overlap %s" % (by / 1e3, inner / 1e9, overlap_rule(by, inner)))


if __name__ == "__main__":
    main()
