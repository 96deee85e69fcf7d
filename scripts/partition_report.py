#!/usr/bin/env python3
"""The partition of a bench configuration over N ranks, computed on the HOST (no GPU): what every rank would own, hold as
ghosts, whom it would exchange with and how many bytes per exchange -- the plan `bench.py --gpus N` executes.

    python scripts/partition_report.py cfg4 --world 8 > profiles/r03_partition_plan_cfg4_8ranks.json

(Eight processes cannot share the one GPU of the build box -- the pool allows six -- so the 8-way split of the full-size
configuration is recorded from the partitioner itself; the 4- and 6-rank runs through tests/mock_rccl execute the same code.)"""
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
    ap.add_argument("--min-dofs", type=int, default=400000)
    args = ap.parse_args()
    import bench
    from alfi_amd import dist as D
    t0 = time.time()
    lv, tr, k = bench.build_problem(args.config, False, lazy=bench.CONFIGS[args.config][0] != "sv")
    t_gen = time.time() - t0
    world = args.world
    splits = D.choose_splits(lv, world, args.min_dofs)
    t0 = time.time()
    ghosts = [[D.compute_ghosts(lv, tr, splits, l, r) for r in range(world)] for l in range(len(lv))]
    t_ghost = time.time() - t0
    levels = []
    for l, L in enumerate(lv):
        s = splits[l]
        distributed = bool(np.count_nonzero(np.diff(s)) > 1)
        row = {"level": l, "dofs": int(L.n), "distributed": distributed}
        if not distributed:
            row["owner"] = 0
            levels.append(row)
            continue
        bs = L.bs
        per = []
        for r in range(world):
            g = ghosts[l][r]
            owner = np.searchsorted(s, g, side="right") - 1
            recv = np.bincount(owner, minlength=world)
            # what r sends to q = q's ghosts owned by r
            send = np.array([np.count_nonzero((ghosts[l][q] >= s[r]) & (ghosts[l][q] < s[r + 1])) if q != r else 0
                             for q in range(world)])
            nbr = np.flatnonzero((recv > 0) | (send > 0))
            lo, hi = int(s[r]), int(s[r + 1])
            npatch = int(len(D.owned_patches(L, lo, hi))) if l > 0 else 0
            per.append({"rank": r, "owned_dofs": int((hi - lo) * bs), "ghost_dofs": int(len(g) * bs),
                        "ghost_fraction": round(len(g) / max(hi - lo, 1), 4), "patches": npatch,
                        "neighbours": [int(q) for q in nbr], "n_neighbours": int(len(nbr)),
                        "forward_halo_KB_sent": round(8e-3 * bs * int(send.sum()), 1),
                        "forward_halo_KB_received": round(8e-3 * bs * int(recv.sum()), 1),
                        "largest_message_KB": round(8e-3 * bs * int(max(send.max(), recv.max())), 1)})
        row["ranks"] = per
        row["neighbours_max"] = max(p["n_neighbours"] for p in per)
        row["owned_dofs_min_max"] = [min(p["owned_dofs"] for p in per), max(p["owned_dofs"] for p in per)]
        row["ghost_fraction_max"] = max(p["ghost_fraction"] for p in per)
        levels.append(row)
    out = {"config": args.config, "workload": bench.describe(args.config), "world": world, "min_dofs": args.min_dofs,
           "host_generation_s": round(t_gen, 1), "ghost_lists_all_ranks_s": round(t_ghost, 1),
           "neighbours_max": max([lv_["neighbours_max"] for lv_ in levels if lv_["distributed"]] + [0]),
           "single_owner_levels": [lv_["level"] for lv_ in levels if not lv_["distributed"]],
           "single_owner_dofs": int(sum(lv_["dofs"] for lv_ in levels if not lv_["distributed"])),
           "levels": levels,
           "note": "exchange counts per V-cycle do not depend on the number of ranks (two distributed levels at config 4: 91 "
                   "halo exchanges + 44 all-reduces, profiles/r02_bench_dist4_native_transport_mock_sharedgpu_functional.json); "
                   "a forward halo exchange is one grouped ncclSend/ncclRecv with exactly the listed neighbours"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
