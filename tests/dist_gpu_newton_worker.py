"""Worker of tests/test_gpu_dist.py::test_partitioned_newton: one rank of the Newton / Reynolds-continuation loop with all
device work on partitioned levels (alfi_amd.dist.DistNavierStokesSolver), ranks sharing the box's single GPU over gloo."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out = sys.argv[1]
    disc = sys.argv[2] if len(sys.argv) > 2 else "pkp0"
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from alfi_amd.dist import DistNavierStokesSolver
    from alfi_amd.nssolver import run_solver
    from alfi_amd.problem import TwoDimLidDrivenCavityProblem
    s = DistNavierStokesSolver(TwoDimLidDrivenCavityProblem(4 if disc == "sv" else 8), 2 if disc == "sv" else 1, 2, min_dofs=1,
                               discretisation=disc)
    res = run_solver(s, [10, 100])
    if rank == 0:
        np.savez(os.path.join(out, "newton.npz"), u=s.u, p=s.p, its=[res[r]["linear_iter"] for r in (10, 100)],
                 newton=[res[r]["nonlinear_iter"] for r in (10, 100)], conv=[res[r]["converged"] for r in (10, 100)])
    s.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
