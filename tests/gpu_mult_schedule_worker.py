"""Worker of tests/test_gpu_env_variants.py::test_persistent_sweep_equals_the_launch_per_wavefront_schedule: symmetrised
multiplicative sweeps (two '|'-separated sort orders, so patches appear twice) on a 2-D [P2]^2, a 3-D [P2+FB]^3 and a 3-D
Scott-Vogelius [P3]^3 macro-star level (patches of more than 64 nodes: the workgroup-per-patch sweep), applied three times each; saves the results to argv[1] (npz).  The parent runs it under ALFI_MULT_PERSISTENT=1 and =0."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from alfi_amd import hip
    from alfi_amd.problem import TwoDimLidDrivenCavityProblem, ThreeDimLidDrivenCavityProblem, build_hierarchy
    from alfi_amd.relaxation import OrderedRelaxation, Options
    ctx = hip.Context(0)
    out = {}
    for name, prob, nref, order in (("2d", TwoDimLidDrivenCavityProblem(8), 2, "0+:1-|1+:0-"),
                                    ("3d", ThreeDimLidDrivenCavityProblem(2), 2, "0+:1-:2+"),
                                    ("sv3d", ThreeDimLidDrivenCavityProblem(1), 1, "0+:1-:2+")):
        if name == "sv3d":      # macro stars of the Scott-Vogelius [P3]^3 pair: several hundred dofs, a WORKGROUP per patch
            from alfi_amd.sv import build_sv_hierarchy
            lv, _ = build_sv_hierarchy(prob, nref, 3, Re=100.0, gamma=1e4)
        else:
            lv, _ = build_hierarchy(prob, nref, 2, Re=100.0)
        L = lv[-1]
        dl = hip.Level(ctx, L.A, L.bc_dofs)
        dl.set_patches(L.patch_ptr, L.patch_dofs)
        dl.factor()
        orl = OrderedRelaxation()
        orl.name = "Star"
        orl.opts = Options("", {"pc_patch_construction_Star_sort_order": order})
        iterset = orl.iteration_order(L.V.mesh.coords[L.patch_seeds])
        nw = dl.set_multiplicative(iterset, True)
        x = np.random.default_rng(7).standard_normal(L.n)
        dx, dy = ctx.vec(x), ctx.vec(L.n)
        ys = []
        for _ in range(3):
            dl.patch_apply(dx, dy)
            ys.append(dy.get())
        assert np.array_equal(ys[0], ys[1]) and np.array_equal(ys[0], ys[2]), "the sweep is not reproducible run to run"
        out[name] = ys[0]
        out[name + "_waves"] = np.int64(nw)
        out[name + "_items"] = np.int64(len(iterset))
        dl.close()
    np.savez(sys.argv[1], **out)
    print("OK", {k: (int(v) if v.ndim == 0 else v.shape) for k, v in out.items()}, flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
