"""sha256 of implicit + explicit half-epochs at k <= 64 (csrc/als_chol.hip) -- two builds that must
be bit-identical print the same lines (LK_AMD_LIBRARY=... python tools/chol_hash.py)."""
import hashlib, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, scipy.sparse as sps, torch
from lkpy_amd import _device as D, _native
rng = np.random.default_rng(11)
LENS = [2049, 3000, 5121, 1, 7, 100, 2048, 0, 130, 333, 1000] + list(rng.integers(1, 400, 500))
n_cols = 20000
indptr = np.zeros(len(LENS) + 1, np.int64); np.cumsum(LENS, out=indptr[1:])
indices = np.concatenate([np.sort(rng.choice(n_cols, n, replace=False)) for n in LENS]).astype(np.int32)
values = rng.integers(1, 6, indptr[-1]).astype(np.float32)
mat = sps.csr_array((values, indices, indptr), shape=(len(LENS), n_cols))
dev = torch.device("cuda:0")
for k in (64, 50, 32, 10):
    other = (rng.standard_normal((n_cols, k)) * 0.05).astype(np.float32)
    csr = D.DeviceCSR.from_arrays(mat.indptr.astype(np.int32), mat.indices, mat.data, mat.shape, dev)
    d_other = D.to_device_padded(other, dev)
    d_otor = D.Gramian(k, dev)(d_other, 0.1)
    plan = D.ALSPlan(csr, k, _native.SOLVER_CHOLESKY)
    d_this = torch.zeros((mat.shape[0], d_other.shape[1]), device=dev)
    plan.half_epoch(d_this, d_other, d_otor); plan.check_status()
    h1 = hashlib.sha256(d_this.cpu().numpy().tobytes()).hexdigest()[:16]
    d_this.zero_()
    plan.half_epoch_explicit(d_this, d_other, 0.1); plan.check_status()
    h2 = hashlib.sha256(d_this.cpu().numpy().tobytes()).hexdigest()[:16]
    print("k", k, "implicit", h1, "explicit", h2, "finite", bool(torch.isfinite(d_this).all()))
