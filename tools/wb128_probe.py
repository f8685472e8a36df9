import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, scipy.sparse as sps, torch
from lkpy_amd import _device as D, _native
rng = np.random.default_rng(0)
n_rows, n_cols, k = 6000, 4000, 256
lens = np.where(rng.random(n_rows) < 0.8, rng.integers(1, 17, n_rows), rng.integers(65, 129, n_rows))
indptr = np.zeros(n_rows + 1, np.int64); np.cumsum(lens, out=indptr[1:])
indices = np.concatenate([np.sort(rng.choice(n_cols, l, replace=False)) for l in lens]).astype(np.int32)
mat = sps.csr_array((np.full(indptr[-1], 40.0, np.float32), indices, indptr), shape=(n_rows, n_cols))
other = (rng.standard_normal((n_cols, k)) * 0.1).astype(np.float32)
dev = torch.device("cuda:0")
csr = D.DeviceCSR.from_arrays(mat.indptr.astype(np.int32), mat.indices, mat.data, mat.shape, dev)
d_other = D.to_device_padded(other, dev)
d_otor = D.Gramian(k, dev)(d_other, 0.1)
res = {}
for mode in ("1", "0"):
    os.environ["LK_ALS_WB128"] = mode
    plan = D.ALSPlan(csr, k, _native.SOLVER_CHOLESKY)
    d_this = torch.zeros((n_rows, 256), device=dev)
    plan.half_epoch(d_this, d_other, d_otor); plan.check_status()
    res[mode] = d_this.cpu().numpy()
    print(mode, "use_wb", plan.use_wb, "short", plan.short_rows, "wb rows", plan.woodbury_rows)
big = lens > 64
print("rows 65..128 differ bitwise:", int((res["1"][big] != res["0"][big]).any(axis=1).sum()), "of", int(big.sum()))
print("rel diff", float(np.linalg.norm(res["1"][big] - res["0"][big]) / np.linalg.norm(res["0"][big])))
