"""Worker of tests/test_gpu_dist.py::test_partitioned_multiplicative: one rank of a partitioned HIP multigrid with
multiplicative (symmetrised) patch sweeps on every smoothed level, compared in place with the SPMD NumPy oracle running
on the same rank-local data through the same Comm (gloo).  Writes <out>/rank<r>.ok on success."""
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
    from alfi_amd.dist import DistMultigrid
    from oracle.dist_oracle import DistOracle
    lv, tr, k, min_dofs = _hier(case)
    dmg = DistMultigrid(lv, tr, k, robust_restriction=False, min_dofs=min_dofs)
    orders = []
    for LL, dl in zip(dmg.local_levels, dmg.levels):
        npatch = len(LL.patch_ptr) - 1
        if LL.level > 0 and npatch > 0:
            order = np.arange(npatch)[::-1].copy()          # any fixed order; the oracle uses the same one
            with torch.cuda.stream(dmg.stream):
                assert dl.set_multiplicative(order, True) >= 1
            orders.append(order)
        else:
            orders.append(None if LL.level == 0 or npatch == 0 else np.zeros(0, dtype=np.int64))
    omg = DistOracle(dmg.local_levels, dmg.local_transfers, dmg.lmin, k, dmg.comm, robust=False, mult_orders=orders,
                     symmetrise=True)
    b = np.random.default_rng(0).standard_normal(lv[-1].n)
    b[lv[-1].bc_dofs] = 0.0
    F = dmg.fine
    bl = np.zeros(F.n)
    bl[:F.n_own] = b[F.part.own_dofs()]
    # one smoother application, then one V-cycle
    top = len(dmg.local_levels) - 1
    db, dx = dmg.local_vec(b), dmg.local_vec()
    with torch.cuda.stream(dmg.stream):
        dmg.levels[-1].patch_apply(db, dx)
    ref = omg.patch_apply(top, bl.copy())
    got = dmg.owned(dx)
    e1 = np.abs(got - ref[:F.n_own]).max() / np.abs(ref[:F.n_own]).max()
    dx = dmg.local_vec()
    dmg.vcycle(db, dx)
    refv = omg.vcycle(top, bl.copy(), np.zeros(F.n))
    gotv = dmg.owned(dx)
    e2 = np.abs(gotv - refv[:F.n_own]).max() / np.abs(refv[:F.n_own]).max()
    assert e1 < 1e-7 and e2 < 1e-5, (e1, e2)
    open(os.path.join(out, "rank%d.ok" % rank), "w").write("%g %g\n" % (e1, e2))
    dmg.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
