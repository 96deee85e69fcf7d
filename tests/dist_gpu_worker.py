"""Worker of tests/test_gpu_dist.py: one rank of a partitioned HIP multigrid, several ranks sharing the box's single GPU
(gloo backend: the halo buffers are staged through the host by alfi_amd.dist.Comm; the library-side exchange sequence is
the one the RCCL path uses).  Writes the rank's owned pieces of two V-cycles and one full cycle to <out>/rank<r>.npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    case, robust, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    import torch
    import torch.distributed as dist
    from tests.test_dist_cpu import _hier
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from alfi_amd.dist import DistMultigrid
    lv, tr, k, min_dofs = _hier(case)
    dmg = DistMultigrid(lv, tr, k, robust_restriction=bool(robust), min_dofs=min_dofs)
    if os.environ.get("ALFI_TEST_EXPECT_TRANSPORT"):
        assert dmg.transport == os.environ["ALFI_TEST_EXPECT_TRANSPORT"], dmg.transport
    if len(sys.argv) > 4:
        # the level smoother alone on a given right-hand side: FGMRES(ks) on the finest level, zero guess
        ks, bfile = int(sys.argv[4]), sys.argv[5]
        b = np.load(bfile)
        db, dx = dmg.local_vec(b), dmg.local_vec()
        with torch.cuda.stream(dmg.stream):
            dmg._in_cycle = True
            dmg.levels[-1].smooth(ks, db, dx, nonzero_guess=False)
            dmg._in_cycle = False
        np.savez(os.path.join(out, "rank%d.npz" % rank), dofs=dmg.fine.part.own_dofs(), xs=dmg.owned(dx))
        dmg.close()
        dist.barrier()
        dist.destroy_process_group()
        return
    b = np.random.default_rng(0).standard_normal(lv[-1].n)
    b[lv[-1].bc_dofs] = 0.0
    db, dx = dmg.local_vec(b), dmg.local_vec()
    dmg.vcycle(db, dx)
    dmg.vcycle(db, dx)
    xv = dmg.owned(dx)
    dmg.fcycle(db, dx)
    xf = dmg.owned(dx)
    p, bs = dmg.fine.part, dmg.fine.bs
    np.savez(os.path.join(out, "rank%d.npz" % rank), dofs=p.own_dofs(), xv=xv, xf=xf)
    dmg.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
