#!/usr/bin/env python3
"""BASELINE config 5 end to end at one-GPU size: bfs3d channel, Scott-Vogelius [P3]^3 - P2dg on the barycentric hierarchy,
Newton with Reynolds continuation (examples/bfs3d/bfs3d.py:49-55), every linear solve on the GPU.

  python scripts/run_cfg5_newton.py [--mesh file.msh] [nref] [Re ...]

--mesh: a gmsh 2.2 ASCII channel (the reference's ``--mesh``, bfs3d.py:13-16), e.g. data/meshes/bfs3d_coarse60.msh;
default: the structured stand-in."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alfi_amd.nssolver import HipNavierStokesSolver, run_solver, performance_info
from alfi_amd.problem import ThreeDimBackwardsFacingStepProblem

argv = sys.argv[1:]
msh = None
if argv and argv[0] == "--mesh":
    msh, argv = argv[1], argv[2:]
nref = int(argv[0]) if argv else 1
res = [float(r) for r in argv[1:]] or [1.0, 10.0, 100.0]
t0 = time.time()
s = HipNavierStokesSolver(ThreeDimBackwardsFacingStepProblem(1, msh=msh), nref, 3, discretisation="sv", verbose=True)
print("mesh: %s" % (msh or "structured stand-in channel"), flush=True)
print("setup %.1f s; velocity dofs %d, pressure dofs %d, macro stars on the finest level %d"
      % (time.time() - t0, s.n_u, s.n_p, len(s.levels[-1].patch_ptr) - 1), flush=True)
s.ctx.prof_enable(True)
results = run_solver(s, res)
print("%8s %8s %8s %10s" % ("Re", "Newton", "Krylov", "minutes"))
for re in res:
    i = results[re]
    print("%8g %8d %8d %10.2f   converged=%s" % (re, i["nonlinear_iter"], i["linear_iter"], i["time"], i["converged"]))
performance_info(s)
s.close()
