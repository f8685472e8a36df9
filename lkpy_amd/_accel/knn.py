"""``lenskit._accel.knn`` stand-in (src/lenskit/_accel/knn.pyi:8-29)."""
from __future__ import annotations

import numpy as np
import torch

from .. import _device as D
from ..data import SparseRowArray
from ..parallel import AccelTask
from ._util import as_csr_arrays


def compute_similarities(ui_ratings, iu_ratings, shape, min_sim: float,
                         save_nbrs: int | None) -> AccelTask:
    """
    Item-item similarity build (src/accel/knn/item_train.rs:33-152).  Returns a task
    yielding a LIST with one ``SparseRowArray`` chunk (int64 offsets, like the reference's
    ``LargeListArray`` chunks; rows in item order, sorted by column).
    """
    nu, ni = shape
    uo, uidx, uval, ushape = as_csr_arrays(ui_ratings)
    io, iidx, ival, ishape = as_csr_arrays(iu_ratings)
    assert ushape == (nu, ni) and ishape == (ni, nu)  # item_train.rs:51-54

    def run(task: AccelTask):
        dev = D.device()
        dt = np.int64 if (uo.dtype == np.int64 or io.dtype == np.int64) else np.int32
        ui = D.DeviceCSR.from_arrays(uo.astype(dt), uidx, uval, ushape, dev)
        iu = D.DeviceCSR.from_arrays(io.astype(dt), iidx, ival, ishape, dev)
        out = D.iknn_build(ui, iu, min_sim, save_nbrs)
        task.set_progress(ni)
        return [SparseRowArray(out.indptr.cpu().numpy(), out.indices.cpu().numpy(),
                               out.values.cpu().numpy(), (ni, ni))]

    return AccelTask(run, total=ni)


def _score(sims, ref_items, ref_rates, tgt_items, max_nbrs, min_nbrs):
    so, sidx, sval, sshape = as_csr_arrays(sims)
    assert sshape[0] == sshape[1]
    dev = D.device()
    dsims = D.DeviceCSR(torch.from_numpy(np.asarray(so, dtype=np.int64)).to(dev),
                        torch.from_numpy(sidx).to(dev), torch.from_numpy(sval).to(dev),
                        sshape, None)

    def nullable(a):  # Arrow arrays with nulls or plain integer arrays (negative = null)
        if hasattr(a, "to_numpy") and hasattr(a, "null_count"):
            vals = a.fill_null(-1).to_numpy(zero_copy_only=False) if a.null_count else \
                a.to_numpy(zero_copy_only=False)
            return np.asarray(vals)
        return np.asarray(a)

    ri = nullable(ref_items).astype(np.int32)
    ti = nullable(tgt_items).astype(np.int32)
    rr = None
    if ref_rates is not None:
        rr_np = ref_rates.fill_null(0).to_numpy(zero_copy_only=False) \
            if hasattr(ref_rates, "fill_null") else np.asarray(ref_rates)
        rr = torch.from_numpy(np.ascontiguousarray(rr_np, dtype=np.float32)).to(dev)
    one = lambda n: torch.tensor([0, n], dtype=torch.int64, device=dev)  # noqa: E731
    s, c = D.iknn_score_batch(dsims, one(len(ri)), torch.from_numpy(ri).to(dev), rr,
                              one(len(ti)), torch.from_numpy(ti).to(dev), max_nbrs, min_nbrs)
    return s.cpu().numpy(), c.cpu().numpy()


def score_explicit(sims, ref_items, ref_rates, tgt_items, max_nbrs: int, min_nbrs: int):
    "(scores f32 with NaN for null, counts i32 with -1 for null targets) -- item_score.rs:23-69."
    return _score(sims, ref_items, ref_rates, tgt_items, max_nbrs, min_nbrs)


def score_implicit(sims, ref_items, tgt_items, max_nbrs: int, min_nbrs: int):
    "item_score.rs:72-111."
    return _score(sims, ref_items, None, tgt_items, max_nbrs, min_nbrs)
