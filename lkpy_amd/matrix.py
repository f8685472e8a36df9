"""
Arrow sparse-row arrays: the wire format of the function seam.

Mirror of the Arrow extension types of ``lenskit.data.matrix``
(src/lenskit/data/matrix.py:35-37,104-560) and of their Rust views
(src/accel/sparse/csr.rs:44-223, src/accel/sparse/consumer.rs:96-142), with the SAME extension
names, storage types and accessors, so that arrays produced here are consumed unchanged by
the reference's callers (``pa.chunked_array(smat).combine_chunks()`` ->
``SparseRowArray.from_array``, src/lenskit/knn/item.py:173-177) and arrays produced by the
reference are accepted here:

* ``lenskit.sparse_index``       ``int32`` column index carrying the row dimension,
* ``lenskit.sparse_row``         ``(Large)List<Struct{index: sparse_index, value: T}>``,
* ``lenskit.sparse_index_list``  ``(Large)List<sparse_index>`` (structure only).

Pure Arrow / NumPy plumbing: nothing here computes.  If the real ``lenskit`` has already
registered the extension names in this process, its classes are used instead.
"""

from __future__ import annotations

import json
from typing import Any

import numpy as np
import pyarrow as pa
import scipy.sparse as sps

SPARSE_IDX_EXT_NAME = "lenskit.sparse_index"
SPARSE_IDX_LIST_EXT_NAME = "lenskit.sparse_index_list"
SPARSE_ROW_EXT_NAME = "lenskit.sparse_row"


class SparseIndexType(pa.ExtensionType):
    """
    Wire schema (category: interchange format, src/lenskit/data/matrix.py:104-143): ``int32``
    storage under the name ``lenskit.sparse_index``; the column count travels as the JSON
    metadata ``{"dimension": n}``.
    """

    def __init__(self, dimension: int):
        self.dimension = int(dimension)
        super().__init__(pa.int32(), SPARSE_IDX_EXT_NAME)

    def check_dimension(self, expected: int | None) -> int:
        "the column count, after confirming it is ``expected`` when the caller states one"
        if expected is None or int(expected) == self.dimension:
            return self.dimension
        raise ValueError(f"sparse index declares {self.dimension} columns, caller expects {expected}")

    def __arrow_ext_serialize__(self) -> bytes:
        return json.dumps({"dimension": self.dimension}).encode()

    @classmethod
    def __arrow_ext_deserialize__(cls, storage_type, serialized):
        if storage_type != pa.int32():
            raise TypeError(f"{SPARSE_IDX_EXT_NAME} is stored as int32, not {storage_type}")
        return cls(json.loads(serialized)["dimension"])

    def __reduce__(self):
        return SparseIndexType, (self.dimension,)


class _RowLayout:
    """
    What one look at an Arrow type tells about a sparse-row array: how many columns, whether
    offsets are 64-bit, and the value type (``None`` = structure only).  Both extension types
    below are thin labels around this description; ``read`` is the single place that decides
    whether a storage type is an acceptable sparse-row layout (the checks the Rust view makes in
    ``CSRMatrix::from_arrow``, src/accel/sparse/csr.rs:160-204, and ``from_type`` of
    src/lenskit/data/matrix.py:175-199,255-293, stated once).
    """

    __slots__ = ("dimension", "large", "value_type")

    def __init__(self, dimension: int, large: bool, value_type):
        self.dimension, self.large, self.value_type = int(dimension), bool(large), value_type

    @staticmethod
    def _columns(index_type: pa.DataType, stated: int | None) -> int:
        if isinstance(index_type, SparseIndexType):
            return index_type.check_dimension(stated)
        if index_type != pa.int32():
            raise TypeError(f"column numbers must be int32 (or {SPARSE_IDX_EXT_NAME}), "
                            f"found {index_type}")
        if stated is None:  # plain int32 indices (pre-extension files) carry no column count
            raise TypeError("plain int32 column numbers need the dimension from the caller")
        return int(stated)

    @classmethod
    def read(cls, data_type: pa.DataType, stated: int | None, want_values: bool) -> "_RowLayout":
        if pa.types.is_large_list(data_type):
            large = True
        elif pa.types.is_list(data_type):
            large = False
        else:
            raise TypeError(f"sparse rows are List or LargeList arrays, found {data_type}")
        elem = data_type.value_type
        if not want_values:
            return cls(cls._columns(elem, stated), large, None)
        if not pa.types.is_struct(elem):
            raise TypeError(f"row elements must be Struct{{index, value}}, found {elem}")
        names = [elem.field(i).name for i in range(elem.num_fields)]
        if names != ["index", "value"]:
            raise TypeError(f"row element fields must be ['index', 'value'], found {names}")
        return cls(cls._columns(elem.field(0).type, stated), large, elem.field(1).type)

    def storage(self, index_type: SparseIndexType) -> pa.DataType:
        elem = index_type if self.value_type is None else \
            pa.struct([("index", index_type), ("value", self.value_type)])
        return pa.large_list(elem) if self.large else pa.list_(elem)


class _SparseRowsBase(pa.ExtensionType):
    "shared behaviour of the two row types: empty metadata, layout recovered from the storage"

    _with_values: bool
    _ext_name: str

    def _setup(self, layout: _RowLayout):
        self.index_type = SparseIndexType(layout.dimension)
        self.value_type = layout.value_type
        pa.ExtensionType.__init__(self, layout.storage(self.index_type), self._ext_name)

    @classmethod
    def from_type(cls, data_type: pa.DataType, dimension: int | None = None):
        if isinstance(data_type, cls):
            data_type.index_type.check_dimension(dimension)
            return data_type
        lay = _RowLayout.read(data_type, dimension, cls._with_values)
        return cls._from_layout(lay)

    @property
    def dimension(self) -> int:
        return self.index_type.dimension

    @property
    def large(self) -> bool:
        return pa.types.is_large_list(self.storage_type)

    def __arrow_ext_serialize__(self) -> bytes:
        return b""  # wire schema: everything is in the storage type

    @classmethod
    def __arrow_ext_deserialize__(cls, storage_type, serialized):
        return cls.from_type(storage_type)

    def __arrow_ext_class__(self):
        return SparseRowArray


class SparseIndexListType(_SparseRowsBase):
    "``lenskit.sparse_index_list``: ``(Large)List<sparse_index>``, structure only."

    _with_values = False
    _ext_name = SPARSE_IDX_LIST_EXT_NAME

    def __init__(self, dimension: int, large: bool = False):
        self._setup(_RowLayout(dimension, large, None))

    @classmethod
    def _from_layout(cls, lay: _RowLayout):
        return cls(lay.dimension, lay.large)

    def __reduce__(self):
        return SparseIndexListType, (self.dimension, self.large)


class SparseRowType(_SparseRowsBase):
    "``lenskit.sparse_row``: ``(Large)List<Struct{index: sparse_index, value: T}>``."

    _with_values = True
    _ext_name = SPARSE_ROW_EXT_NAME

    def __init__(self, dimension: int, value_type: pa.DataType | None = pa.float32(),
                 large: bool = False):
        self._setup(_RowLayout(dimension, large, value_type))

    @classmethod
    def _from_layout(cls, lay: _RowLayout):
        return cls(lay.dimension, lay.value_type, lay.large)

    def __reduce__(self):
        return SparseRowType, (self.dimension, self.value_type, self.large)


class SparseRowArray(pa.ExtensionArray):
    """
    An array of sparse rows = a CSR matrix (matrix.py:318-539): ``offsets`` (int32, or int64
    for ``LargeList``), ``indices`` (int32) and ``values`` are zero-copy Arrow views of the
    three CSR buffers -- exactly what the HIP kernels read (``include/lkamd.h``, "CSR").
    """

    @classmethod
    def from_arrays(cls, offsets, indices, values=None, *, shape=None) -> "SparseRowArray":
        offsets = pa.array(offsets)
        large = pa.types.is_int64(offsets.type)
        if isinstance(indices, pa.ExtensionArray):
            indices = indices.storage
        indices = pa.array(indices, type=pa.int32())
        if shape:
            _nr, nc = shape
        else:
            import pyarrow.compute as pc

            nc = pc.max(indices).as_py() + 1
        if values is not None:
            values = pa.array(values)
            row_type = SparseRowType(nc, values.type, large)
            idx = pa.ExtensionArray.from_storage(row_type.index_type, indices)
            # non-nullable fields, like the Rust side builds them (consumer.rs:98-103)
            fields = [pa.field("index", row_type.index_type), pa.field("value", values.type)]
            elements = pa.StructArray.from_arrays([idx, values], fields=fields)
        else:
            row_type = SparseIndexListType(nc, large)
            elements = pa.ExtensionArray.from_storage(row_type.index_type, indices)
        if large:
            storage = pa.LargeListArray.from_arrays(offsets, elements)
        else:
            storage = pa.ListArray.from_arrays(offsets, elements)
        return pa.ExtensionArray.from_storage(row_type, storage.cast(row_type.storage_type))

    @classmethod
    def from_array(cls, array: pa.Array, dimension: int | None = None) -> "SparseRowArray":
        "Interpret an Arrow array as sparse rows; legacy layouts included (matrix.py:364-385)."
        if isinstance(array, pa.ChunkedArray):
            array = array.combine_chunks()
            if isinstance(array, pa.ChunkedArray):  # pyarrow >= 15 may keep one chunk
                array = array.chunk(0) if array.num_chunks == 1 else pa.concat_arrays(array.chunks)
        if isinstance(array, SparseRowArray):
            array.type.index_type.check_dimension(dimension)
            return array
        if isinstance(array.type, (SparseRowType, SparseIndexListType)):
            array.type.index_type.check_dimension(dimension)
            return pa.ExtensionArray.from_storage(array.type, array)
        elem = getattr(array.type, "value_type", None)
        structure_only = elem is not None and not pa.types.is_struct(elem)
        et = (SparseIndexListType if structure_only else SparseRowType).from_type(
            array.type, dimension)
        return pa.ExtensionArray.from_storage(et, array.cast(et.storage_type))

    @classmethod
    def from_scipy(cls, matrix, *, values: bool = True, large: bool | None = None):
        "matrix.py:387-424: int32 offsets unless nnz >= 2^31 or ``large=True``."
        matrix = sps.csr_array(matrix)
        matrix.sort_indices()
        smax = np.iinfo(np.int32).max
        offsets = matrix.indptr
        if large:
            offsets = np.require(offsets, np.int64)
        elif matrix.nnz < smax:
            offsets = np.require(offsets, dtype=np.int32)
        elif large is False:
            raise ValueError(f"sparse matrix size {matrix.nnz:,d} too large for list")
        vals = pa.array(matrix.data) if values else None
        return cls.from_arrays(offsets, np.require(matrix.indices, np.int32), vals,
                               shape=matrix.shape)

    def to_scipy(self) -> sps.csr_array:
        if not self.has_values:
            raise TypeError("structure-only arrays cannot convert to scipy")
        return sps.csr_array(
            (self.values.to_numpy(), self.indices.to_numpy(), self.offsets.to_numpy()),
            shape=(len(self), self.type.dimension),
        )

    @property
    def dimension(self) -> int:
        return self.type.dimension

    @property
    def shape(self) -> tuple[int, int]:
        return (len(self), self.dimension)

    @property
    def has_values(self) -> bool:
        return self.type.value_type is not None

    @property
    def offsets(self) -> pa.Array:
        return self.storage.offsets

    @property
    def indices(self) -> pa.Int32Array:
        vals = self.storage.values
        idx = vals.field(0) if self.has_values else vals
        return idx.storage if isinstance(idx, pa.ExtensionArray) else idx

    @property
    def values(self) -> pa.Array | None:
        return self.storage.values.field(1) if self.has_values else None

    @property
    def nnz(self) -> int:
        return self.offsets[len(self)].as_py()

    def structure(self) -> "SparseRowArray":
        if self.has_values:
            return self.from_arrays(self.offsets, self.indices, shape=self.shape)
        return self

    def transpose(self) -> "SparseRowArray":
        "matrix.py:512-530, through the device transpose (``lk_csr_transpose``)."
        from ._accel import data as _data_accel

        nr, nc = self.shape
        row_ptr, col_ind, perm = _data_accel.transpose_csr(self.structure(), self.has_values)
        if perm is None:
            return self.from_arrays(row_ptr, col_ind, shape=(nc, nr))
        return self.from_arrays(row_ptr, col_ind, self.values.take(perm), shape=(nc, nr))

    def row_extent(self, row: int) -> tuple[int, int]:
        return self.storage.offsets[row].as_py(), self.storage.offsets[row + 1].as_py()

    def row_indices(self, row: int) -> pa.Int32Array:
        sp, ep = self.row_extent(row)
        return self.indices.slice(sp, ep - sp)

    def row_values(self, row: int):
        if not self.has_values:
            return None
        sp, ep = self.row_extent(row)
        return self.values.slice(sp, ep - sp)


def _register():
    for t in (SparseIndexType(0), SparseIndexListType(0), SparseRowType(0)):
        try:
            pa.register_extension_type(t)
        except pa.ArrowKeyError:
            pass  # the real lenskit (or an earlier import) owns the name already


_register()


def csr_arrays(matrix: Any):
    """
    Anything the reference's boundary accepts as a CSR matrix -> NumPy views
    ``(offsets, indices, values | None, (rows, cols))`` without copying the buffers:
    a :class:`SparseRowArray` (ours or ``lenskit.data.matrix``'s), a raw Arrow
    ``ListArray`` / ``LargeListArray`` of ``Struct{index, value}`` or of ``int32`` carrying the
    ``lenskit.sparse_index`` extension (what the Rust side receives through the C Data
    Interface, src/accel/sparse/csr.rs:160-204), or a SciPy sparse matrix.  Wrong types raise
    ``TypeError`` like ``CSRMatrix::from_arrow``.
    """
    if sps.issparse(matrix):
        matrix = SparseRowArray.from_scipy(matrix)
    if isinstance(matrix, pa.ChunkedArray):
        matrix = matrix.combine_chunks()
    if not isinstance(matrix, pa.Array):
        raise TypeError(f"invalid array type {type(matrix)}, expected List or LargeList")
    try:
        sra = SparseRowArray.from_array(matrix)
    except TypeError as e:
        raise TypeError(f"invalid array type: {e}") from e
    if sra.null_count:
        raise TypeError("sparse row arrays cannot contain null rows")
    n = len(sra)
    off = sra.offsets.to_numpy(zero_copy_only=False)
    # a sliced array keeps the parent's buffers: rebase
    base = int(off[0]) if n else 0
    idx = sra.indices.to_numpy(zero_copy_only=False)
    vals = None if not sra.has_values else sra.values.to_numpy(zero_copy_only=False)
    if base:
        end = int(off[-1])
        off = off - off[0]
        idx = idx[base:end]
        vals = None if vals is None else vals[base:end]
    return off, idx, vals, (n, sra.dimension)


def fast_col_cooc(rows, cols, shape, *, progress=None, include_diagonal: bool = True,
                  ordered: bool = False, dense: bool = False):
    """
    Column co-occurrence counts ``M^T M`` of a binary matrix given as COO coordinates --
    ``lenskit.data.matrix.fast_col_cooc`` (src/lenskit/data/matrix.py:599-646) over
    ``_accel.data.count_cooc`` / ``dense_cooc`` (src/accel/data/cooc.rs:47-192, symmetric pair
    counter src/accel/data/pairs/symmetric.rs).  On the device this IS the item-item
    accumulation of the similarity build with unit values: every shared group adds 1.0f to the
    pair's cell (exact integers up to 2^24), cells >= 0.5 are kept; the diagonal (the column
    counts) is added on request.  Returns a ``coo_array`` of int32 counts (both triangles), or a
    dense int32 array with ``dense=True``.  ``ordered=True`` (position-ordered pair counts
    inside a group) is not part of the similarity path and is not offered.
    """
    import torch  # noqa: F401

    from . import _device as D

    if ordered:
        raise NotImplementedError("ordered co-occurrence counts are not offered by the device "
                                  "path (only the symmetric M^T M form)")
    m, n = shape
    rows = np.asarray(rows if not isinstance(rows, pa.Array) else rows.to_numpy(), dtype=np.int32)
    cols = np.asarray(cols if not isinstance(cols, pa.Array) else cols.to_numpy(), dtype=np.int32)
    if len(rows) != len(cols):
        raise ValueError("array length mismatch")  # cooc.rs:60-62
    ones = np.ones(len(rows), dtype=np.float32)
    ui = sps.csr_array((ones, (rows, cols)), shape=(m, n))
    ui.sum_duplicates()
    ui.data[:] = 1.0  # a binary matrix: an (group, item) pair counts once
    ui.sort_indices()
    iu = sps.csr_array(ui.T)
    iu.sort_indices()
    dev = D.device()
    out = D.iknn_build(D.DeviceCSR.from_scipy(ui, dev), D.DeviceCSR.from_scipy(iu, dev), 0.5)
    ptr = out.indptr.cpu().numpy()
    r = np.repeat(np.arange(n, dtype=np.int32), np.diff(ptr))
    c = out.indices.cpu().numpy()
    v = np.rint(out.values.cpu().numpy()).astype(np.int32)
    if include_diagonal:
        cnt = np.diff(iu.indptr).astype(np.int32)
        nz = np.flatnonzero(cnt).astype(np.int32)
        r, c, v = np.concatenate([r, nz]), np.concatenate([c, nz]), np.concatenate([v, cnt[nz]])
    res = sps.coo_array((v, (r, c)), shape=(n, n))
    return res.toarray() if dense else res
