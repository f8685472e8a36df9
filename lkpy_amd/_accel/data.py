"""``lenskit._accel.data`` stand-ins on the path: ``argtopn`` / ``argsort_descending``
(src/accel/data/sorting.rs:69-172) and ``transpose_csr`` (src/accel/data/transpose.rs:19-108)."""
from __future__ import annotations

import numpy as np
import torch

from .. import _device as D

_TOPN_MAX = 4096


def _scores(scores) -> np.ndarray:
    if hasattr(scores, "to_numpy") and hasattr(scores, "null_count"):  # pyarrow
        arr = scores.to_numpy(zero_copy_only=False).astype(np.float32, copy=True)
        if scores.null_count:
            import pyarrow.compute as pc

            arr[pc.is_null(scores).to_numpy(zero_copy_only=False)] = np.nan
        return arr
    return np.ascontiguousarray(scores, dtype=np.float32)


def argtopn(scores, n: int) -> np.ndarray:
    """
    Indices of the ``n`` largest non-NaN / non-null scores, descending (sorting.rs:132-172);
    ``n <= 0`` gives an empty result like the Rust.  Ties: lower index first.
    """
    s = _scores(scores)
    if n <= 0 or len(s) == 0:
        return np.empty(0, dtype=np.int32)
    n = min(int(n), len(s))
    if n > _TOPN_MAX:
        raise ValueError(f"argtopn: n={n} exceeds the kernel limit {_TOPN_MAX}")
    dev = D.device()
    out = D.argtopn(torch.from_numpy(s).to(dev).unsqueeze(0), n)[0].cpu().numpy()
    return out[out >= 0]


def argsort_descending(scores) -> np.ndarray:
    "All valid indices by descending score (sorting.rs:69-103)."
    s = _scores(scores)
    valid = int(np.sum(~np.isnan(s)))
    if valid == 0:
        return np.empty(0, dtype=np.int32)
    return argtopn(s, valid)


def transpose_csr(matrix, permute: bool):
    """
    ``transpose_csr(structure, permute)`` (src/lenskit/_accel/data.pyi:12,
    src/accel/data/transpose.rs:19-108): (row offsets, column indices, permutation | None) of
    the transposed structure, same offset width as the input; stable (entries of an output
    row in input order).
    """
    from ._util import as_csr_arrays

    offsets, indices, _vals, shape = as_csr_arrays(matrix)
    dev = D.device()
    csr = D.DeviceCSR.from_arrays(offsets, indices, np.zeros(len(indices), np.float32), shape, dev)
    t = D.csr_transpose(csr, with_values=bool(permute))
    perm = t.perm.cpu().numpy() if permute else None
    return t.indptr.cpu().numpy(), t.indices.cpu().numpy(), perm
