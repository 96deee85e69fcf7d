#!/usr/bin/env python3
"""MFMA-busy fraction per kernel from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` run:
busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), summed over the launches of a kernel (the
convention of profiles/r01f_cfg5s_mfma_busy.json).  usage: mfma_busy_summary.py <pmc_dir> <out.json> <what>"""
import collections
import csv
import glob
import json
import re
import sys


def main():
    d, out, what = sys.argv[1:4]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
            k = re.sub(r"^void ", "", re.sub(r"\(.*", "", k))
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k].add(r["Dispatch_Id"])
    res = {}
    for k, v in agg.items():
        mf, gui = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("GRBM_GUI_ACTIVE", 0.0)
        if mf <= 0 or gui <= 0:
            continue
        res[k] = {"launches": len(launches[k]), "SQ_VALU_MFMA_BUSY_CYCLES": mf, "GRBM_GUI_ACTIVE": gui,
                  "mfma_busy_fraction": mf / (gui / 8.0 * 1024.0)}
    json.dump({"what": what, "note": "MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), summed over the "
               "launches of a kernel; kernels without MFMA instructions omitted", "kernels": dict(sorted(res.items()))},
              open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["SQ_VALU_MFMA_BUSY_CYCLES"]):
        print("%-50s launches %5d  busy %.3f" % (k[:50], v["launches"], v["mfma_busy_fraction"]))


if __name__ == "__main__":
    main()
