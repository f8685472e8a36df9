"""A/B of als_wb4_kernel against als_wb_kernel per row length (rows of 0..8 entries)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, scipy.sparse as sps, torch
from lkpy_amd import _device as D, _native
rng = np.random.default_rng(0)
n_rows, n_cols = 4000, 4000
k = int(os.environ.get("K", "256"))
lens = rng.integers(0, 13, n_rows)
indptr = np.zeros(n_rows + 1, np.int64); np.cumsum(lens, out=indptr[1:])
indices = np.concatenate([np.sort(rng.choice(n_cols, l, replace=False)) for l in lens]).astype(np.int32)
mat = sps.csr_array((np.full(indptr[-1], 40.0, np.float32), indices, indptr), shape=(n_rows, n_cols))
other = (rng.standard_normal((n_cols, k)) * 0.1).astype(np.float32)
dev = torch.device("cuda:0")
csr = D.DeviceCSR.from_arrays(mat.indptr.astype(np.int32), mat.indices, mat.data, mat.shape, dev)
d_other = D.to_device_padded(other, dev)
d_otor = D.Gramian(k, dev)(d_other, 0.1)
res = {}
os.environ["LK_ALS_WB_MIN_ROWS"] = "1"
for mode in ("1", "0"):
    os.environ["LK_ALS_WB4"] = mode; os.environ["LK_ALS_WB8"] = mode
    plan = D.ALSPlan(csr, k, _native.SOLVER_CHOLESKY)
    d_this = torch.zeros((n_rows, d_other.shape[1]), device=dev)
    plan.half_epoch(d_this, d_other, d_otor); plan.check_status()
    res[mode] = d_this.cpu().numpy()
    print(mode, "use_wb", plan.use_wb, "short", plan.short_rows)
order = np.argsort(-lens, kind="stable")
pos = np.empty(n_rows, np.int64); pos[order] = np.arange(n_rows)
t4 = int((lens > 4).sum())
for n in range(0, 13):
    m = lens == n
    a, b = res["1"][m], res["0"][m]
    err = np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-6)
    print("n", n, "rows", int(m.sum()), "max rel", float(err.max()), "bad", int((err > 1e-3).sum()))
bad = np.flatnonzero(np.linalg.norm(res["1"] - res["0"], axis=1) > 1e-3 * np.maximum(np.linalg.norm(res["0"], axis=1), 1e-6))
print("bad rows:", bad[:20], "len", lens[bad[:20]], "slot in wave", (pos[bad[:20]] - t4) % 4)
for r in bad[:3]:
    a, b = res["1"][r], res["0"][r]
    q = a.shape[0] // 4
    print("row", r, "quarter errs", [float(np.linalg.norm(a[i*q:(i+1)*q] - b[i*q:(i+1)*q])) for i in range(4)],
          "ratio", float(np.dot(a, b) / np.dot(b, b)))
G = other.astype(np.float64).T @ other.astype(np.float64) + 0.1 * np.eye(k)
Zf = np.linalg.solve(G, other.astype(np.float64).T).T
for r in list(bad[:8]):
    cols = mat.indices[mat.indptr[r]:mat.indptr[r + 1]]
    zz = Zf[cols]
    ga = np.linalg.lstsq(zz.T, res["1"][r, :k].astype(np.float64), rcond=None)[0]
    gb = np.linalg.lstsq(zz.T, res["0"][r, :k].astype(np.float64), rcond=None)[0]
    S0 = other[cols].astype(np.float64) @ zz.T
    print("row", r, "n", len(cols), "slot", int((pos[r] - t4) % 4), "g wb4", np.round(ga, 4), "g ref", np.round(gb, 4),
          "w", 41.0, "diag S0", np.round(np.diag(S0), 4))
