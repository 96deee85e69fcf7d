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
    if disc == "pkp0-3d":        # [P2+FB]^3 - P0, the element of BASELINE config 4
        from alfi_amd.problem import ThreeDimLidDrivenCavityProblem
        s = DistNavierStokesSolver(ThreeDimLidDrivenCavityProblem(2), 1, 2, min_dofs=1)
    elif disc == "pkp0-supg":    # the authors' production set-up: SUPG stabilisation, its terms assembled on the device as well
        s = DistNavierStokesSolver(TwoDimLidDrivenCavityProblem(8), 1, 2, min_dofs=1, stabilisation_type="supg")
    else:
        s = DistNavierStokesSolver(TwoDimLidDrivenCavityProblem(4 if disc == "sv" else 8), 2 if disc == "sv" else 1, 2,
                                   min_dofs=1, discretisation=disc)
    # every host assembly during the Newton loops is counted (operator rows and SUPG terms): the device path must need none
    from alfi_amd import _hostlib
    calls = []
    real, real_supg = _hostlib.assemble_bsr, _hostlib.supg
    _hostlib.assemble_bsr = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    _hostlib.supg = lambda *a, **k: (calls.append(1), real_supg(*a, **k))[1]
    # what crosses PCIe per Newton step of the SECOND solve (alfi_transfer_stats: every copy the library makes): with the state
    # distributed on the devices only scalars
    res = {10: s.solve(10)[1]}
    s.ctx.transfer_stats(reset=True)
    res[100] = s.solve(100)[1]
    h2d, d2h = s.ctx.transfer_stats()
    per_step = max(h2d, d2h) / max(res[100]["nonlinear_iter"], 1)
    resident = bool(s._device_state_resident())
    u_all, p_all = s.u, s.p                      # COLLECTIVE: every rank contributes its owned entries
    _hostlib.assemble_bsr, _hostlib.supg = real, real_supg
    asm_err = -1.0
    if s.device_assembly:
        # the operator values the device assembled from the final state against the rank-local host assembly of the same rows
        from alfi_amd.dist import localize_operator
        from alfi_amd.lazy import LazyOperator
        asm_err = 0.0
        s._upload_states(s.u)
        for asm, dl, LL, w in zip(s._asm, s.dmg.levels, s.dmg.local_levels, s._winds(s.u)[s.dmg.lmin:]):
            if asm is None:
                continue
            L = s.levels[LL.level]
            dl.assemble(s.nu, s.gamma, 1.0, asm[1], True)
            ref = localize_operator(LazyOperator(L.V, L.A.rowptr, L.A.colidx, L.V.mesh.cell_geometry(),
                                                 L.V.element.reference_tensors(), s.nu, s.gamma, 1.0, np.ascontiguousarray(w),
                                                 full_div=s.sv),
                                    LL.part)
            asm_err = max(asm_err, float(np.abs(dl.get_values() - ref.vals).max() / np.abs(ref.vals).max()))
    gathered = [None] * world
    dist.all_gather_object(gathered, (len(calls), asm_err, bool(s.device_assembly), per_step, resident))
    if rank == 0:
        np.savez(os.path.join(out, "newton.npz"), u=u_all, p=p_all, its=[res[r]["linear_iter"] for r in (10, 100)],
                 bytes_per_step=[g[3] for g in gathered], resident=[g[4] for g in gathered],
                 newton=[res[r]["nonlinear_iter"] for r in (10, 100)], conv=[res[r]["converged"] for r in (10, 100)],
                 host_assemblies=[g[0] for g in gathered], asm_err=[g[1] for g in gathered],
                 device_assembly=[g[2] for g in gathered])
    s.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
