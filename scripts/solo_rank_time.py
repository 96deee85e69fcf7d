#!/usr/bin/env python3
"""Per-rank COMPUTE time of an N-GPU run, measured on ONE GPU: every rank of the N-way mesh partition is set up alone in this
process (alfi_amd.dist.DistMultigrid(solo=(rank, N)): the partition, the localisation, the patch factors and every kernel launch are
that rank's; the exchange points do nothing) and its V-cycles are timed by the library's device events per class.  The values such
a run computes are meaningless (ghosts are never filled) -- the device time of the kernels is what the rank would spend computing.
What this does NOT measure: the exchanges themselves (profiles/r05_overlap_rule.txt has their fixed cost over real RCCL), link
bandwidth, skew between ranks.

  python scripts/solo_rank_time.py cfg4 --world 8 [--ranks 0 3 7] [--cycles 5]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--ranks", type=int, nargs="*", default=None)
    ap.add_argument("--cycles", type=int, default=5)
    ap.add_argument("--min-dofs", type=int, default=400000)
    ap.add_argument("--cycle", choices=["v", "f"], default="v", help="time V-cycles, or full cycles (pc_mg_type full: what the outer "
                    "solves apply) with the partition balanced for them")
    args = ap.parse_args()
    import torch
    import bench
    from alfi_amd.dist import DistMultigrid
    torch.cuda.set_device(0)
    t0 = time.time()
    lv, tr, k = bench.build_problem(args.config, False, lazy=True)
    print("%s: rank-local generation %.1f s; %d levels, finest %d dofs" % (args.config, time.time() - t0, len(lv), lv[-1].n), flush=True)
    L = lv[-1]
    b = np.random.default_rng(0).standard_normal(L.n)
    b[L.bc_dofs] = 0.0
    rows = []
    for world, ranks in ((1, [0]), (args.world, args.ranks if args.ranks else list(range(args.world)))):
        for r in ranks:
            t0 = time.time()
            dmg = DistMultigrid(lv, tr, k, solo=(r, world), min_dofs=args.min_dofs, full_cycle=args.cycle == "f")
            cycle = dmg.fcycle if args.cycle == "f" else dmg.vcycle
            dmg.sync()
            t_setup = time.time() - t0
            db, dx = dmg.local_vec(b), dmg.local_vec()
            for _ in range(2):
                cycle(db, dx)
            dmg.sync()
            dmg.ctx.prof_enable(True)
            dmg.ctx.prof_reset()
            t0 = time.time()
            for _ in range(args.cycles):
                cycle(db, dx)
            dmg.sync()
            wall = 1e3 * (time.time() - t0) / args.cycles
            prof = dmg.ctx.prof_get()
            dmg.ctx.prof_enable(False)
            ev = {name: prof[name][0] / args.cycles for name in prof}
            compute = sum(v for name, v in ev.items() if name != "COMM")
            own = [int(p.nb_own) * p.bs for p in dmg.parts]
            row = {"world": world, "rank": r, "compute_ms": round(compute, 3), "wall_ms_with_host_callbacks": round(wall, 2),
                   "setup_s": round(t_setup, 1), "owned_dofs_by_level": own,
                   "events_ms": {name: round(v, 3) for name, v in ev.items()}}
            rows.append(row)
            print(json.dumps(row), flush=True)
            dmg.close()
            del dmg
            torch.cuda.empty_cache()
    one = rows[0]["compute_ms"]
    many = [x["compute_ms"] for x in rows[1:]]
    print("device time of the kernels per %s-cycle: 1 rank %.2f ms; %d ranks: max %.2f / mean %.2f / min %.2f ms "
          "-> compute-only speed-up %.2f (ideal %d); sum over ranks / 1 rank = %.3f (ghost redundancy + small-launch inefficiency)"
          % ("F" if args.cycle == "f" else "V", one, args.world, max(many), sum(many) / len(many), min(many), one / max(many), args.world,
             sum(many) / one if len(many) == args.world else float("nan")))


if __name__ == "__main__":
    main()
