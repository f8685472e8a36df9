from __future__ import annotations

import numpy as np
import scipy.sparse as sps

from ..data import SparseRowArray


def as_csr_arrays(matrix):
    "SparseRowArray (ours or the reference's Arrow one), or SciPy -> (offsets, indices, values, shape)."
    if isinstance(matrix, SparseRowArray):
        return matrix.offsets, matrix.indices, matrix.values, matrix.shape
    if sps.issparse(matrix):
        m = SparseRowArray.from_scipy(matrix)
        return m.offsets, m.indices, m.values, m.shape
    if hasattr(matrix, "offsets") and hasattr(matrix, "indices"):  # lenskit.data.matrix.SparseRowArray
        def np_(a):
            return a.to_numpy(zero_copy_only=False) if hasattr(a, "to_numpy") else np.asarray(a)

        vals = matrix.values
        return np_(matrix.offsets), np_(matrix.indices), None if vals is None else np_(vals), matrix.shape
    raise TypeError(f"expected a SparseRowArray, got {type(matrix)}")  # csr.rs:161-193 -> TypeError
