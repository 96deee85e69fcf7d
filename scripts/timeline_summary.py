#!/usr/bin/env python3
"""Attribute ONE traced V-cycle (scripts/trace_cycle.py under rocprofv3 --kernel-trace --memory-copy-trace) to kernel
time, memory copies and the idle gaps between consecutive device activities.

usage: timeline_summary.py <results.db> [cycles=4]

Cycle boundaries = idle gaps of more than 1 ms on the device (trace_cycle.py sleeps 2 ms on the host in front of every
cycle and synchronises after it).  For the last ``cycles`` cycles: wall span first-start .. last-end, number of activities,
summed kernel time, summed copy time, summed gaps, the gap histogram, and a table per (kernel, grid) with launches,
kernel time and the gap in FRONT of those launches -- so the sum of the table is the span."""
import collections
import re
import sqlite3
import sys

import numpy as np


def tables(db):
    return [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]


def main():
    db = sqlite3.connect(sys.argv[1])
    cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    tb = tables(db)
    kd = [t for t in tb if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tb if t.startswith("rocpd_info_kernel_symbol")][0]
    acts = []
    for name, gx, wx, s, e in db.execute("select s.kernel_name, d.grid_size_x, d.workgroup_size_x, d.start, d.end from %s d join %s s "
                                         "on d.kernel_id = s.id" % (kd, ks)):
        n = re.sub(r"\(.*", "", name)
        n = re.sub(r"^void ", "", n)
        acts.append((s, e, "K", n[:60], int(gx) // max(int(wx), 1)))
    mc = [t for t in tb if t.startswith("rocpd_memory_copy")]
    ncopy = 0
    if mc:
        cols = [r[1] for r in db.execute("pragma table_info(%s)" % mc[0])]
        size_col = "size" if "size" in cols else None
        q = "select start, end%s from %s" % ((", " + size_col) if size_col else "", mc[0])
        for row in db.execute(q):
            acts.append((row[0], row[1], "C", "memory copy (%s B)" % (row[2] if size_col else "?"), 0))
            ncopy += 1
    acts.sort()
    start = np.array([a[0] for a in acts], dtype=np.int64)
    end = np.array([a[1] for a in acts], dtype=np.int64)
    # running end (activities may overlap when two streams are involved)
    run_end = np.maximum.accumulate(end)
    gap = np.concatenate([[0], np.maximum(start[1:] - run_end[:-1], 0)]) / 1e3
    bounds = [i for i in range(len(acts)) if gap[i] > 1000.0]
    if len(bounds) < cycles:
        raise SystemExit("only %d cycle boundaries in the trace" % len(bounds))
    segs = list(zip(bounds[-cycles:], bounds[-cycles + 1:] + [len(acts)])) if cycles > 1 else [(bounds[-1], len(acts))]
    print("%d activities in the trace (%d memory copies), %d cycle boundaries; last %d cycles:" % (len(acts), ncopy, len(bounds), len(segs)))
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    hist = np.zeros(7, dtype=np.int64)
    edges = [0.5, 1, 2, 3, 5, 10]
    tot_span = tot_k = tot_c = tot_g = 0.0
    nact = 0
    for i0, i1 in segs:
        span = (run_end[i1 - 1] - start[i0]) / 1e3
        d = (end[i0:i1] - start[i0:i1]) / 1e3
        g = gap[i0:i1].copy()
        g[0] = 0.0
        kinds = [a[2] for a in acts[i0:i1]]
        ksum = sum(x for x, kk in zip(d, kinds) if kk == "K")
        csum = sum(x for x, kk in zip(d, kinds) if kk == "C")
        print("  span %9.1f us  activities %4d  kernel %9.1f us  copies %7.1f us  gaps %8.1f us  (kernel+copy+gaps = %.1f)"
              % (span, i1 - i0, ksum, csum, g.sum(), ksum + csum + g.sum()))
        tot_span += span; tot_k += ksum; tot_c += csum; tot_g += g.sum(); nact += i1 - i0
        for a, dd, gg in zip(acts[i0:i1], d, g):
            v = agg[(a[3], a[4])]
            v[0] += 1; v[1] += dd; v[2] += gg
        hist += np.histogram(g[1:], bins=[-1] + edges + [1e9])[0]
    nc = len(segs)
    print("per cycle: span %.1f us, %d activities, kernel %.1f us, copies %.1f us, gaps %.1f us (%.2f us per boundary)"
          % (tot_span / nc, nact // nc, tot_k / nc, tot_c / nc, tot_g / nc, tot_g / max(nact - nc, 1)))
    print("gap histogram (us) per cycle: " + ", ".join("%s%s: %.0f" % ("<=" if i < 6 else ">", edges[min(i, 5)], hist[i] / nc)
                                                         for i in range(7)))
    print("%-62s %8s %7s %10s %8s %10s" % ("activity", "wgs", "n/cyc", "us/cycle", "avg us", "gap us/cyc"))
    rows = sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))
    for k, v in rows[:70]:
        print("%-62s %8d %7.1f %10.1f %8.2f %10.1f" % (k[0], k[1], v[0] / nc, v[1] / nc, v[1] / v[0], v[2] / nc))
    # by workgroup-count class: where do the gaps sit?
    cls = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for k, v in agg.items():
        c = "<=64 wgs" if k[1] <= 64 else "<=1024 wgs" if k[1] <= 1024 else "<=16384 wgs" if k[1] <= 16384 else "> 16384 wgs"
        cls[c][0] += v[0]; cls[c][1] += v[1]; cls[c][2] += v[2]
    print("by launch size (per cycle):")
    for c in ("<=64 wgs", "<=1024 wgs", "<=16384 wgs", "> 16384 wgs"):
        v = cls[c]
        print("  %-12s launches %6.1f  kernel %9.1f us  gaps in front %8.1f us" % (c, v[0] / nc, v[1] / nc, v[2] / nc))


if __name__ == "__main__":
    main()
