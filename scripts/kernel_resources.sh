#!/bin/bash
# Register / scratch / occupancy table of the kernels of one .hip file (compile-time remarks of the gfx950 backend).
#   scripts/kernel_resources.sh alfi_amd/csrc/kernels_assemble.hip [name filter]
src=$1; filt=${2:-.}
root=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -I $root/include -I $root/alfi_amd/csrc -c $src -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys, re, subprocess
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark: [^ ]* +Function Name: (\S+)|Name: (\S+)", line)
    if m and "Function Name" in line:
        cur = {"name": m.group(1) or m.group(2)}; rows.append(cur); continue
    for key in ("TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m and cur is not None and key not in cur: cur[key] = int(m.group(1))
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n")
print("%-70s %5s %5s %5s %7s %4s %6s" % ("kernel", "SGPR", "VGPR", "AGPR", "scratch", "occ", "LDS"))
for r, n in zip(rows, names):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"\(.*", "", n); n = n.replace("void ", "")
    print("%-70s %5d %5d %5d %7d %4d %6d" % (n[:70], r.get("TotalSGPRs", 0), r.get("VGPRs", 0), r.get("AGPRs", 0), r.get("ScratchSize [bytes/lane]", 0), r.get("Occupancy [waves/SIMD]", 0), r.get("LDS Size [bytes/block]", 0)))
' | grep -E "kernel +SGPR|$filt"
