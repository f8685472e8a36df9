"""``lenskit._accel.data`` stand-ins on the path: ``argtopn`` / ``argsort_descending``
(src/accel/data/sorting.rs:69-172) and ``transpose_csr`` (src/accel/data/transpose.rs:19-108).
Arrow in, Arrow out, like the PyO3 functions."""
from __future__ import annotations

import numpy as np
import pyarrow as pa
import torch

from .. import _device as D


def _scores(scores) -> np.ndarray:
    """
    float / integer Arrow array (nulls allowed), NumPy array or sequence -> float32 with NaN
    for "not a candidate" (null or NaN: sorting.rs:143,78-84).  The kernels compare float32;
    float64 / wide-integer scores are rounded to float32 first (the reference compares them in
    their own type -- scores that differ only beyond float32 precision may order differently).
    """
    if isinstance(scores, pa.ChunkedArray):
        scores = scores.combine_chunks()
    if isinstance(scores, pa.Array):
        if not (pa.types.is_floating(scores.type) or pa.types.is_integer(scores.type)):
            # match_array_type! in sorting.rs -> TypeError
            raise TypeError(f"unsupported score array type {scores.type}")
        arr = scores.fill_null(0).to_numpy(zero_copy_only=False).astype(np.float32, copy=True)
        if scores.null_count:
            arr[scores.is_null().to_numpy(zero_copy_only=False)] = np.nan
        return arr
    return np.ascontiguousarray(scores, dtype=np.float32)


def _i32(arr: np.ndarray) -> pa.Int32Array:
    return pa.array(np.ascontiguousarray(arr, dtype=np.int32), type=pa.int32())


def argtopn(scores, n: int) -> pa.Int32Array:
    """
    Positions of the ``n`` largest non-NaN / non-null scores, descending
    (sorting.rs:132-172); ``n <= 0`` gives an empty result like the Rust.  Ties: lower index
    first (the reference's heap order among equal scores is unspecified).  Any ``n``: lists
    beyond the selection kernel's 4096 entries are fully sorted on the device.
    """
    s = _scores(scores)
    if n is None or n <= 0 or len(s) == 0:
        return _i32(np.empty(0, dtype=np.int32))
    n = min(int(n), len(s))
    dev = D.device()
    out = D.argtopn(torch.from_numpy(s).to(dev).unsqueeze(0), n)[0].cpu().numpy()
    return _i32(out[out >= 0])


def argsort_descending(scores) -> pa.Int32Array:
    "All valid positions by descending score (sorting.rs:69-103)."
    s = _scores(scores)
    if len(s) == 0:
        return _i32(np.empty(0, dtype=np.int32))
    dev = D.device()
    out = D.argtopn(torch.from_numpy(s).to(dev).unsqueeze(0), -1)[0].cpu().numpy()
    return _i32(out[out >= 0])


def transpose_csr(matrix, permute: bool):
    """
    ``transpose_csr(structure, permute)`` (src/lenskit/_accel/data.pyi:12,
    src/accel/data/transpose.rs:19-108): (row offsets, column indices, permutation | None) of
    the transposed structure as Arrow arrays, same offset width as the input; stable (entries
    of an output row in input order).
    """
    from ._util import as_csr_arrays

    offsets, indices, _vals, shape = as_csr_arrays(matrix)
    dev = D.device()
    csr = D.DeviceCSR.from_arrays(offsets, indices, np.zeros(len(indices), np.float32), shape, dev)
    t = D.csr_transpose(csr, with_values=bool(permute))
    perm = pa.array(t.perm.cpu().numpy()) if permute else None
    return pa.array(t.indptr.cpu().numpy()), _i32(t.indices.cpu().numpy()), perm
