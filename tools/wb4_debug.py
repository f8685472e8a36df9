import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, scipy.sparse as sps, torch
from lkpy_amd import _device as D, _native
rng = np.random.default_rng(0)
n_rows, n_cols, k = 64, 4000, 256
lens = rng.integers(1, 5, n_rows)
indptr = np.zeros(n_rows + 1, np.int64); np.cumsum(lens, out=indptr[1:])
indices = np.concatenate([np.sort(rng.choice(n_cols, l, replace=False)) for l in lens]).astype(np.int32)
mat = sps.csr_array((np.full(indptr[-1], 40.0, np.float32), indices, indptr), shape=(n_rows, n_cols))
other = (rng.standard_normal((n_cols, k)) * 0.1).astype(np.float32)
dev = torch.device("cuda:0")
csr = D.DeviceCSR.from_arrays(mat.indptr.astype(np.int32), mat.indices, mat.data, mat.shape, dev)
d_other = D.to_device_padded(other, dev)
d_otor = D.Gramian(k, dev)(d_other, 0.1)
os.environ["LK_ALS_WB_MIN_ROWS"] = "1"
plan = D.ALSPlan(csr, k, _native.SOLVER_CHOLESKY)
d_this = torch.zeros((n_rows, 256), device=dev)
plan.half_epoch(d_this, d_other, d_otor)
x = d_this.cpu().numpy()
order = np.argsort(-lens, kind="stable")
G = other.astype(np.float64).T @ other.astype(np.float64) + 0.1 * np.eye(k)
Zf = np.linalg.solve(G, other.astype(np.float64).T).T
np.set_printoptions(linewidth=200, precision=4, suppress=True)
for wv in range(2):
    rows = order[4 * wv:4 * wv + 4]
    d = x[rows[0]]
    print("wave", wv, "rows", rows, "lens", lens[rows])
    for nm, o in (("u'", 0), ("rhs", 16), ("piv", 32), ("sv", 48), ("w", 64), ("dinv", 80), ("diag", 96)):
        print(" ", nm, d[o:o + 16])
    want_rhs = np.zeros(16); want_diag = np.ones(16)
    for i, r in enumerate(rows):
        cols = mat.indices[mat.indptr[r]:mat.indptr[r + 1]]
        S0 = other[cols].astype(np.float64) @ Zf[cols].T
        n = len(cols)
        want_diag[4 * i:4 * i + n] = 1 + 40 * np.diag(S0)
        want_rhs[4 * i:4 * i + n] = np.sqrt(40) * (S0 @ np.full(n, 41.0))
    print("  want diag", want_diag)
    print("  want rhs ", want_rhs)
