#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) into profiles/pmc_<kernel>_<cfg>.json.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE is reported in KiB and tallies the 128-B requests of a
wide streaming read at 64 B => bytes = 2 * 1024 * FETCH_SIZE; WRITE_SIZE (KiB) is exact for streaming stores.

usage: pmc_summary.py <fetch_dir> <write_dir> <kernel-name-prefix> <out.json> [tag] [first] [grids] [skip]

``skip``: leading launches of the kernel to drop -- since round 2 every patch factorisation ends with a residual probe that
runs the level's apply kernel once (one launch per smoothed level before the first cycle).

``first``: only the first N launches of the kernel -- the V-cycle of ``bench.py --steps 1 --warmup 0`` (60 patch applies, 79
SpMVs on config 4); the full cycles bench.py runs afterwards for ``fcycle_ms`` have a different mix of levels.
``grids``: ``max`` (only the launches on the kernel's largest grid: the finest level) or comma-separated Grid_Size values -- only launches of those sizes (config 5: the smoother's big_apply_kernel launches,
903680 and 310272 threads; the transfers' interior solves use the same kernel on smaller grids).
"""
import glob
import json
import sys

import pandas as pd


def per_kernel(d, counter, prefix, first=None, grids=None, skip=0):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    t = pd.read_csv(f)
    # several kernels that together make one operation ("void a+void b+void c": launched in turn, the same number of
    # times): the counters of the i-th launch of each are added
    # "a+b+c|d+e+f": two such operations (the launches of the first, then of the second): one list of per-operation sums
    if "|" in prefix:
        import numpy as np
        return np.concatenate([per_kernel(d, counter, q, None, grids, 0) for q in prefix.split("|")])[skip:][:first]
    if "+" in prefix:
        parts = [per_kernel(d, counter, q, first, grids, skip) for q in prefix.split("+")]
        m = min(len(q) for q in parts)
        return sum(q[:m] for q in parts)
    t = t[(t["Counter_Name"] == counter) & t["Kernel_Name"].str.startswith(prefix)]
    if grids == "max":         # the launches on the largest grid only (the finest level)
        t = t[t["Grid_Size"] == t["Grid_Size"].max()]
    elif grids:
        t = t[t["Grid_Size"].isin(grids)]
    # one row per dispatch and counter instance: sum the instances, keep dispatch order
    v = t.groupby("Dispatch_Id", sort=True)["Counter_Value"].sum().to_numpy()[skip:]
    return v[:first] if first else v


def provenance():
    """What the summary was measured ON: the commit the caller names (ALFI_COMMIT: the GPU box holds a snapshot without .git),
    the box, the time, and the SHA-256 of every kernel source -- bench.py quotes a summary as `roofline.traffic` only while
    those files are byte for byte what was measured."""
    import datetime
    import hashlib
    import os
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "alfi_amd", "csrc")
    sha = {}
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            sha[f] = hashlib.sha256(open(os.path.join(csrc, f), "rb").read()).hexdigest()
    return {"commit": os.environ.get("ALFI_COMMIT", "unknown"), "box": socket.gethostname(),
            "collected": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"), "kernel_sources_sha256": sha}


def main():
    fetch_dir, write_dir, prefix, out = sys.argv[1:5]
    tag = sys.argv[5] if len(sys.argv) > 5 else ""
    first = int(sys.argv[6]) if len(sys.argv) > 6 else None
    grids = None
    if len(sys.argv) > 7 and sys.argv[7] not in ("", "-"):
        grids = "max" if sys.argv[7] == "max" else [int(g) for g in sys.argv[7].split(",")]
    skip = int(sys.argv[8]) if len(sys.argv) > 8 else 0
    first = first or None
    f, w = (per_kernel(fetch_dir, "FETCH_SIZE", prefix, first, grids, skip),
            per_kernel(write_dir, "WRITE_SIZE", prefix, first, grids, skip))
    fb, wb = 2.0 * 1024.0 * f.mean(), 1024.0 * w.mean()
    res = {"kernel": prefix, "tag": tag, "launches": int(len(f)), "skipped_leading_launches": skip,
           "fetch_size_KiB_avg_raw": float(f.mean()), "write_size_KiB_avg_raw": float(w.mean()),
           "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb,
           "fetch_bytes_per_launch_max": 2048.0 * float(f.max()),
           "correction": "FETCH_SIZE KiB x 1024 x 2 (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE KiB x 1024"}
    res.update(provenance())
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
