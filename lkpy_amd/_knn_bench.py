"""Item-kNN model-build leg of bench.py (BASELINE.json metric, second half)."""
from __future__ import annotations

import time

import numpy as np
import scipy.sparse as sps
import scipy.sparse.linalg as spla
import torch


def prepare_explicit(ratings: sps.csr_array):
    """
    Host preparation exactly as ``ItemKNNScorer.train`` does it with SciPy
    (src/lenskit/knn/item.py:142-156,202-228): item-mean centring, L2 normalisation.
    """
    rmat = sps.coo_array(ratings).astype(np.float32).tocsc()
    counts = np.diff(rmat.indptr)
    sums = rmat.sum(axis=0)
    means = np.zeros(sums.shape, dtype=np.float32)
    np.divide(sums, counts, out=means, where=counts > 0)
    rmat.data = rmat.data - np.repeat(means, counts)
    norms = spla.norm(rmat, 2, axis=0)
    cmat = (rmat / np.maximum(norms, np.finfo("f4").smallest_normal)).astype(np.float32)
    ui = sps.csr_array(cmat.tocsr())
    iu = sps.csr_array(cmat.T.tocsr())
    ui.sort_indices()
    iu.sort_indices()
    return ui, iu, means


def run(ratings: sps.csr_array, dev, reps: int = 2) -> dict:
    from . import _device as D

    t0 = time.perf_counter()
    ui, iu, _ = prepare_explicit(ratings)
    t_prep = time.perf_counter() - t0
    dui = D.DeviceCSR.from_scipy(ui, dev)
    diu = D.DeviceCSR.from_scipy(iu, dev)
    torch.cuda.synchronize(dev)
    times = []
    out = None
    for _ in range(reps):
        del out
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        out = D.iknn_build(dui, diu, 1.0e-6, None)
        torch.cuda.synchronize(dev)
        times.append(time.perf_counter() - t0)
    macs = int((np.diff(ui.indptr).astype(np.int64) ** 2).sum())
    best = min(times)
    return {
        "metric": "item-kNN model build seconds (ML-25M-shaped, cosine, min_sim=1e-6, unbounded)",
        "value": round(best, 4),
        "unit": "s",
        "higher_is_better": False,
        "build_seconds_all": [round(t, 4) for t in times],
        "host_prepare_seconds": round(t_prep, 3),
        "nnz_out": int(out.indices.shape[0]),
        "macs": macs,
        "gmacs_per_s": round(macs / best / 1e9, 2),
        "note": "one compute pass into an n_items^2 staging area + compaction; CSR resident in HBM, output left in HBM",
    }
