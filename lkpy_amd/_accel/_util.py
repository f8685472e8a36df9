from __future__ import annotations

import numpy as np
import pyarrow as pa

from ..matrix import csr_arrays


def as_csr_arrays(matrix):
    """
    What the reference's native functions accept as a sparse matrix -- a ``SparseRowArray``
    (ours or ``lenskit.data.matrix``'s), a raw Arrow ``ListArray`` / ``LargeListArray`` of
    ``Struct{index, value}`` (src/accel/sparse/csr.rs:160-204) or, for convenience, SciPy --
    as NumPy views (offsets, indices, values | None, shape).  Wrong types -> ``TypeError``.
    """
    return csr_arrays(matrix)


def nullable_i32(a) -> np.ndarray:
    "Arrow Int32 array with nulls (or a plain integer array; negative = null) -> int32, null = -1."
    if isinstance(a, (pa.Array, pa.ChunkedArray)):
        if isinstance(a, pa.ChunkedArray):
            a = a.combine_chunks()
        if not pa.types.is_integer(a.type):
            # checked_array_ref("...", "Int32", ...) in item_score.rs -> TypeError
            raise TypeError(f"invalid item array type {a.type}, expected Int32")
        if a.null_count:
            a = a.fill_null(-1)
        return np.array(a.to_numpy(zero_copy_only=False), dtype=np.int32)  # writable copy
    return np.array(a, dtype=np.int32)


def f32_with_nulls(values: np.ndarray) -> pa.FloatArray:
    "float32 with NaN marking 'no score' -> Arrow Float32 with nulls (accum.rs:190-240)."
    values = np.ascontiguousarray(values, dtype=np.float32)
    return pa.array(values, type=pa.float32(), mask=np.isnan(values))


def i32_with_nulls(values: np.ndarray) -> pa.Int32Array:
    "int32 with -1 marking a null target -> Arrow Int32 with nulls (accum.rs:180-194)."
    values = np.ascontiguousarray(values, dtype=np.int32)
    return pa.array(values, type=pa.int32(), mask=values < 0)
