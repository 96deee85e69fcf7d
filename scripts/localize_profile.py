#!/usr/bin/env python3
"""Where a rank's partition + localisation time goes (host only, no GPU):  python scripts/localize_profile.py cfg4 --world 4 --rank 1"""
import argparse
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--world", type=int, default=4)
    ap.add_argument("--rank", type=int, default=1)
    ap.add_argument("--lines", type=int, default=40)
    args = ap.parse_args()
    import bench
    from alfi_amd import dist
    t0 = time.time()
    lv, tr, k = bench.build_problem(args.config, False, lazy=True)
    print("lazy generation %.1f s" % (time.time() - t0))
    pr = cProfile.Profile()
    pr.enable()
    t0 = time.time()
    splits = dist.choose_splits(lv, args.world)
    parts = dist.build_parts(lv, tr, splits, args.rank)
    t1 = time.time()
    llev, ltr, lmin = dist.localize(lv, tr, parts)
    t2 = time.time()
    pr.disable()
    print("%s rank %d of %d: partition %.1f s (ghost lists of every rank computed here: no all-gather), localize %.1f s"
          % (args.config, args.rank, args.world, t1 - t0, t2 - t1))
    pstats.Stats(pr).sort_stats("cumulative").print_stats(args.lines)


if __name__ == "__main__":
    main()
