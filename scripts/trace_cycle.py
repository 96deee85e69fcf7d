#!/usr/bin/env python3
"""Run NOTHING but V-cycles after the setup, so that the tail of a rocprofv3 trace is pure cycle work:

    rocprofv3 --kernel-trace --memory-copy-trace -d <out> -o run -- python3 scripts/trace_cycle.py --config cfg2 --cycles 6

Setup, two warm-up cycles, a stream synchronisation, then ``--cycles`` V-cycles each followed by a stream synchronisation
(the host gap between them marks the cycle boundaries in the trace), no HIP events, no extra cycles afterwards.
scripts/timeline_summary.py attributes kernel time and gaps of the last cycles.  Prints the wall time per cycle."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--cycles", type=int, default=6)
    ap.add_argument("--restriction", action="store_true")
    args = ap.parse_args()
    import bench
    from alfi_amd import hip
    lv, tr, k = bench.build_problem(args.config, False)
    ctx = hip.Context(0)
    mg = hip.Multigrid(ctx, lv, tr, k, robust_restriction=args.restriction)
    L = lv[-1]
    b = np.random.default_rng(0).standard_normal(L.n)
    b[L.bc_dofs] = 0.0
    db, dx = ctx.vec(b), ctx.vec(L.n)
    for _ in range(2):
        mg.vcycle(db, dx)
    ctx.sync()
    times = []
    for _ in range(args.cycles):
        time.sleep(0.002)                  # a visible host gap in front of every traced cycle
        t0 = time.perf_counter()
        mg.vcycle(db, dx)
        ctx.sync()
        times.append(1e3 * (time.perf_counter() - t0))
    print("trace_cycle %s: %s ms per cycle (wall, sync after every cycle)" % (args.config, ["%.3f" % t for t in times]))
    mg.close()
    ctx.close()


if __name__ == "__main__":
    main()
