import time, sys, numpy as np, torch
import scipy.sparse as sp, scipy.sparse.linalg as spla
sys.path.insert(0, ".")
from alfi_amd.problem import ThreeDimLidDrivenCavityProblem, build_hierarchy
from alfi_amd.hip import dense_inverse_gpu
lv, tr = build_hierarchy(ThreeDimLidDrivenCavityProblem(7), 0, 2, Re=1000.0)
As = lv[0].A.to_scipy().tocsc()
n = As.shape[0]
b = np.random.default_rng(0).standard_normal(n); b[lv[0].bc_dofs] = 0
lu = spla.splu(As); xref = lu.solve(b)
for nb, rf in ((256, 0), (256, 1), (256, 2)):
    t = time.time(); X = dense_inverse_gpu(lv[0].A, nb=nb, refine=rf); torch.cuda.synchronize(); print("nb", nb, "blocked GJ", time.time() - t, flush=True)
    x = (X @ torch.from_numpy(b).cuda()).cpu().numpy()
    print("rel err vs splu", np.abs(x - xref).max() / np.abs(xref).max(), "rel resid", np.linalg.norm(As @ x - b) / np.linalg.norm(b), flush=True)
# small-matrix sanity of torch.linalg.inv
M = torch.from_numpy(As[:3000, :3000].toarray()).cuda()
Mi = torch.linalg.inv(M)
print("torch.linalg.inv 3000 resid", float((M @ Mi - torch.eye(3000, dtype=M.dtype, device="cuda")).abs().max()))
