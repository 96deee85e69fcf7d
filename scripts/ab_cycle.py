"""A/B of library switches on one small case: every variant in a fresh process (the switches are read once), outputs compared."""
import os, subprocess, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np
from tests.test_dist_cpu import _hier
from alfi_amd import hip
case = sys.argv[1]
lv, tr, k, _ = _hier(case)
ctx = hip.Context(0)
mg = hip.Multigrid(ctx, lv, tr, k, robust_restriction=True)
b = np.random.default_rng(0).standard_normal(lv[-1].n); b[lv[-1].bc_dofs] = 0
db, dx = ctx.vec(b), ctx.vec(lv[-1].n)
mg.vcycle(db, dx); mg.vcycle(db, dx)
xv = dx.get()
mg.fcycle(db, dx)
np.savez(sys.argv[2], xv=xv, xf=dx.get())
''' % ROOT
case = sys.argv[1] if len(sys.argv) > 1 else "3d-P2FB"
variants = {"default": {}, "nofusedsm": {"ALFI_FUSED_SMOOTHER": "0"},
            "nofuse": {"ALFI_FUSED_SMOOTHER": "0", "ALFI_FUSED_REDUCE": "0"},
            "noalign": {"ALFI_FUSED_SMOOTHER": "0", "ALFI_SPMV_ALIGNED": "0"},
            "neither": {"ALFI_FUSED_SMOOTHER": "0", "ALFI_FUSED_REDUCE": "0", "ALFI_SPMV_ALIGNED": "0"}}
out = {}
for name, env in variants.items():
    f = "/tmp/ab_%s.npz" % name
    subprocess.run([sys.executable, "-c", WORKER, case, f], env=dict(os.environ, **env), check=True, cwd=ROOT)
    out[name] = np.load(f)
ref = out["neither"]
for name in variants:
    z = out[name]
    print("%-8s vs neither: V %.3e  F %.3e" % (name, np.abs(z["xv"] - ref["xv"]).max() / np.abs(ref["xv"]).max(),
                                               np.abs(z["xf"] - ref["xf"]).max() / np.abs(ref["xf"]).max()))
