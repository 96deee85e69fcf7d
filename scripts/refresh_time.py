#!/usr/bin/env python3
"""The operator refresh and the residual evaluation of a Newton step in isolation, at a bench configuration's size (what
PatchPC.update / the SNES residual cost in the reference, alfi/solver.py:320, 325, 204-234): every level's
alfi_level_assemble(_supg) from a synthetic state, alfi_level_assemble_mult + alfi_level_supg on the finest level.

  python scripts/refresh_time.py cfg4 [--supg 0.05] [--reps 3]

Prints device time per call (HIP events of the library, class PATCH_FACTOR) and the algorithmic bytes of the finest level's
three kernels; run under rocprofv3 --kernel-trace --stats / --pmc for the per-kernel numbers (scripts/gpu_r5f.sh)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--supg", type=float, default=None)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import bench
    from alfi_amd.nssolver import HipNavierStokesSolver
    from alfi_amd.problem import TwoDimLidDrivenCavityProblem, ThreeDimLidDrivenCavityProblem
    dim, baseN, nref, ke, Re, k = bench.CONFIGS[args.config]
    prob = TwoDimLidDrivenCavityProblem(baseN) if dim == 2 else ThreeDimLidDrivenCavityProblem(baseN)
    s = HipNavierStokesSolver(prob, nref, ke, stabilisation_type="supg" if args.supg is not None else None,
                              stabilisation_weight=args.supg)
    s.nu = s.char_L * s.char_U / Re
    ctx = s.ctx
    L = s.levels[-1]
    # a smooth synthetic state with the lid values (the regularised lid profile extended into the domain)
    x = L.V.node_coords
    w = np.zeros((L.V.num_nodes, dim))
    w[:, 0] = (x[:, 0] * (2 - x[:, 0])) ** 2 * x[:, 1] ** 2 / 4
    s._dz.set(np.concatenate([w.ravel(), np.zeros(s.n_p)]))
    mgl = s.hmg.mg.levels
    fin = mgl[-1]
    ncell, nloc = L.V.cell_nodes.shape
    nnzb = L.A.nnzb
    bb = dim * dim
    print("%s: finest level %d cells x %d nodes, %d blocks; element blocks %.2f GB, operator %.2f GB, contributor lists %.2f GB"
          % (args.config, ncell, nloc, nnzb, ncell * nloc * nloc * bb * 8 / 1e9, nnzb * bb * 8 / 1e9,
             (8 * (nnzb + 1) + 6 * ncell * nloc * nloc) / 1e9), flush=True)

    def timed(what, fn):
        fn()
        ctx.sync()
        ctx.prof_enable(True)
        ctx.prof_reset()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            fn()
        ctx.sync()
        wall = 1e3 * (time.perf_counter() - t0) / args.reps
        dev = ctx.prof_get()["PATCH_FACTOR"][0] / args.reps
        ctx.prof_enable(False)
        print("%-62s %8.2f ms device, %8.2f ms wall" % (what, dev, wall), flush=True)

    s._device_states(None)

    def refresh_all():
        for dl, st in zip(mgl, s._dstate):
            if args.supg is not None:
                dl.assemble_supg(s.nu, s.gamma, 1.0, st, s.supg_weight, s.supg_magic, True)
            else:
                dl.assemble(s.nu, s.gamma, 1.0, st, True)
    timed("refresh of all %d levels%s" % (len(mgl), " with SUPG" if args.supg is not None else ""), refresh_all)
    st = s._dstate[-1]
    if args.supg is not None:
        timed("finest level: assemble_supg", lambda: fin.assemble_supg(s.nu, s.gamma, 1.0, st, s.supg_weight, s.supg_magic, True))
    timed("finest level: assemble", lambda: fin.assemble(s.nu, s.gamma, 1.0, st, True))
    Fu = s._dres
    timed("finest level: matrix-free product (the residual's A(u) u)", lambda: fin.assemble_mult(s.nu, s.gamma, 0.5, st, st, Fu))
    if args.supg is not None:
        timed("finest level: SUPG residual", lambda: fin.supg(s.nu, s.supg_weight, s.supg_magic, st, False, Fu))
    s.close()


if __name__ == "__main__":
    main()
