#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- builds oracle/_ref/libalfi_ref_bubble.so from the reference's OWN native code.

The only native code in florianwechsung/alfi are the five C kernels it hands to PyOP2 as string literals in
alfi/bubble.py (split :57-91, splitadj :93-123, combine :124-147, combineadj :149-175, count :176-186).  This recipe
reads them where they lie (/root/reference/alfi/bubble.py), pipes them to gcc on stdin and writes nothing but the shared
object into oracle/_ref/ (git-ignored; it travels to the GPU box like any other built .so).  No reference source text is
stored in this repository or in oracle/_ref/.  When /root/reference is absent (the GPU box) this is a no-op and the
prebuilt library is used.

Everything else on the hot path lives in PETSc / Firedrake / PyOP2, which are not installed: unbuildable here
(DESIGN.md section 2)."""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/alfi/bubble.py"
OUT_DIR = os.path.join(HERE, "_ref")
OUT = os.path.join(OUT_DIR, "libalfi_ref_bubble.so")
NAMES = ["split", "splitadj", "combine", "combineadj", "count"]


def build(force=False):
    if not os.path.exists(REF):
        return OUT if os.path.exists(OUT) else None
    if os.path.exists(OUT) and not force and os.path.getmtime(OUT) >= os.path.getmtime(REF):
        return OUT
    text = open(REF).read()
    kernels = re.findall(r'op2\.Kernel\("""(.*?)""",\s*"(\w+)"\)', text, flags=re.S)
    got = {name: src for src, name in kernels}
    missing = [n for n in NAMES if n not in got]
    if missing:
        raise RuntimeError("kernels %s not found in %s" % (missing, REF))
    os.makedirs(OUT_DIR, exist_ok=True)
    src = "\n".join(got[n] for n in NAMES)
    subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-x", "c", "-", "-o", OUT], input=src.encode(), check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
