"""Worker of tests/test_gpu_dist.py::test_partitioned_outer_solve: one rank of the outer FGMRES / fieldsplit-Schur solve on
partitioned levels (alfi_amd.dist.DistSaddle), several ranks sharing the box's single GPU over gloo.  Writes the rank's
owned velocity / pressure pieces of the solution to <out>/rank<r>.npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    case, out = sys.argv[1], sys.argv[2]
    import torch
    import torch.distributed as dist
    from tests.test_dist_cpu import _hier
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from alfi_amd.dist import DistMultigrid, DistSaddle
    from alfi_amd.problem import build_pressure_coupling
    lv, tr, k, min_dofs = _hier(case)
    L = lv[-1]
    dmg = DistMultigrid(lv, tr, k, robust_restriction=False, min_dofs=min_dofs)
    if "SV" in case:
        from alfi_amd.sv import build_sv_pressure_coupling
        B, _, Minv = build_sv_pressure_coupling(L)
        sad = DistSaddle(dmg, B, None, L.V.cell_nodes, L.nu, L.gamma, remove_constant_nullspace=True, mass_inv=Minv)
    else:
        B, vol = build_pressure_coupling(L)
        sad = DistSaddle(dmg, B, vol, L.V.cell_nodes, L.nu, L.gamma, remove_constant_nullspace=True)
    b = np.random.default_rng(0).standard_normal(L.n)
    b[L.bc_dofs] = 0.0
    p = dmg.fine.part
    rhs = torch.tensor(np.concatenate([b[p.own_dofs()], np.zeros(sad.np_own)]), dtype=torch.float64, device=dmg.device)
    x, its, rn = sad.solve(rhs, rtol=1e-9, atol=1e-12)
    dmg.sync()
    x = x.cpu().numpy()
    np.savez(os.path.join(out, "rank%d.npz" % rank), dofs=p.own_dofs(), cells=sad.cells, xu=x[:sad.n_own], xp=x[sad.n_own:],
             its=its, rn=rn)
    sad.close()
    dmg.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
