#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace run (the rocpd SQLite database it writes) per kernel and grid size.

usage: rocpd_summary.py <results.db> [cycles]

The timed cycles of bench.py are delimited by the coarse solves (dense_gemv_kernel, one per V-cycle): the last ``cycles``
of them (default 10) are summarised -- launches, summed kernel time and summed idle gaps in front of the launches -- so
that kernel time and dependency / launch latency of the small levels can be told apart."""
import collections
import re
import sqlite3
import sys

import numpy as np


def main():
    db = sqlite3.connect(sys.argv[1])
    cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    c = db.cursor()
    rows = list(c.execute("select s.kernel_name, d.grid_size_x, d.start, d.end from rocpd_kernel_dispatch d "
                          "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"))
    names = [re.sub(r"\(.*", "", r[0]) for r in rows]
    names = [re.sub(r"^void ", "", n)[:56] for n in names]
    grid = np.array([r[1] for r in rows])
    start = np.array([r[2] for r in rows], dtype=np.int64)
    end = np.array([r[3] for r in rows], dtype=np.int64)
    idx = [i for i, n in enumerate(names) if "dense_gemv" in n]
    if len(idx) <= cycles:
        raise SystemExit("fewer than %d coarse solves in the trace" % (cycles + 1))
    i0, i1 = idx[-cycles - 1], idx[-1]
    dur = (end - start)[i0:i1] / 1e3
    gap = np.maximum((start[1:] - end[:-1])[i0 - 1:i1 - 1] / 1e3, 0.0)
    total = (end[i1] - end[i0]) / 1e3
    print("%d cycles: %.1f us per cycle, %d launches per cycle, kernel time %.1f us, gaps %.1f us per cycle"
          % (cycles, total / cycles, (i1 - i0) // cycles, dur.sum() / cycles, gap.sum() / cycles))
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for n, g, d, gp in zip(names[i0:i1], grid[i0:i1], dur, gap):
        a = agg[(n, int(g))]
        a[0] += 1
        a[1] += d
        a[2] += gp
    print("%-58s %9s %7s %12s %9s %12s" % ("kernel", "grid", "n/cyc", "us/cycle", "avg us", "gap us/cyc"))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-58s %9d %7.1f %12.1f %9.2f %12.1f" % (k[0], k[1], v[0] / cycles, v[1] / cycles, v[1] / v[0], v[2] / cycles))


if __name__ == "__main__":
    main()
