"""Arrow sparse-row wire format (lkpy_amd/matrix.py) -- host-only checks mirroring
tests/data/test_arrow_sparse.py of the reference: same extension names, storage types,
accessors and error behaviour as ``lenskit.data.matrix`` (src/lenskit/data/matrix.py:104-560)
and acceptance of what the Rust side emits (src/accel/sparse/consumer.rs:96-130)."""
import pickle

import numpy as np
import pyarrow as pa
import pytest
import scipy.sparse as sps

from lkpy_amd.matrix import (SPARSE_IDX_EXT_NAME, SPARSE_ROW_EXT_NAME, SparseIndexListType,
                             SparseIndexType, SparseRowArray, SparseRowType, csr_arrays)


def _mat(seed=0, shape=(50, 30), density=0.2):
    m = sps.random_array(shape, density=density, format="csr", dtype=np.float32,
                         rng=np.random.default_rng(seed))
    m.sort_indices()
    return m


def test_extension_names_and_storage():
    a = SparseRowArray.from_scipy(_mat())
    assert isinstance(a, SparseRowArray) and isinstance(a.type, SparseRowType)
    assert a.type.extension_name == SPARSE_ROW_EXT_NAME == "lenskit.sparse_row"
    st = a.type.storage_type
    assert pa.types.is_list(st) and pa.types.is_struct(st.value_type)
    idx_f, val_f = st.value_type.field(0), st.value_type.field(1)
    assert idx_f.name == "index" and val_f.name == "value" and val_f.type == pa.float32()
    assert isinstance(idx_f.type, SparseIndexType) and idx_f.type.dimension == 30
    assert idx_f.type.extension_name == SPARSE_IDX_EXT_NAME
    assert a.shape == (50, 30) and a.dimension == 30 and a.has_values
    big = SparseRowArray.from_scipy(_mat(), large=True)
    assert pa.types.is_large_list(big.type.storage_type)
    assert big.offsets.type == pa.int64() and a.offsets.type == pa.int32()


def test_round_trip_and_accessors():
    m = _mat(1)
    a = SparseRowArray.from_scipy(m)
    assert a.nnz == m.nnz
    assert np.array_equal(a.offsets.to_numpy(), m.indptr)
    assert np.array_equal(a.indices.to_numpy(), m.indices)
    assert np.array_equal(a.values.to_numpy(), m.data)
    assert (a.to_scipy() != m).nnz == 0
    s, e = a.row_extent(7)
    assert (s, e) == (m.indptr[7], m.indptr[8])
    assert np.array_equal(a.row_indices(7).to_numpy(), m.indices[s:e])
    st = a.structure()
    assert isinstance(st.type, SparseIndexListType) and not st.has_values and st.values is None
    assert np.array_equal(st.indices.to_numpy(), m.indices)
    with pytest.raises(TypeError):
        st.to_scipy()
    p = pickle.loads(pickle.dumps(a))
    assert isinstance(p, SparseRowArray) and p.equals(a)


def _rust_chunk(m):
    "what ArrowCSRConsumer::complete emits (consumer.rs:96-130)"
    it = SparseIndexType(m.shape[1])
    idx = pa.ExtensionArray.from_storage(it, pa.array(m.indices, pa.int32()))
    fields = [pa.field("index", it, nullable=False), pa.field("value", pa.float32(), nullable=False)]
    rows = pa.StructArray.from_arrays([idx, pa.array(m.data)], fields=fields)
    lt = pa.large_list(pa.field("rows", rows.type, nullable=False))
    return pa.LargeListArray.from_arrays(pa.array(m.indptr.astype(np.int64)), rows, type=lt)


def test_reference_consumer_lines_on_rust_shaped_chunks():
    "src/lenskit/knn/item.py:173-197, line for line, on chunks shaped like the Rust output"
    m1, m2 = _mat(2, (20, 20)), _mat(3, (15, 20))
    smat = [_rust_chunk(m1), _rust_chunk(m2)]
    assert isinstance(smat, list)
    smat = pa.chunked_array(smat)
    smat = smat.combine_chunks()
    assert pa.types.is_large_list(smat.type)
    smat = SparseRowArray.from_array(smat)
    lengths = np.diff(smat.offsets)
    assert np.sum(lengths > 0) > 0
    assert smat.offsets[-1].as_py() == len(smat.values)
    item_counts = np.diff(smat.offsets.to_numpy())
    want = sps.vstack([m1, m2]).tocsr()
    assert np.array_equal(item_counts, np.diff(want.indptr)) and smat.shape == (35, 20)
    assert np.array_equal(smat.values.to_numpy(), want.data)


def test_csr_arrays_accepts_what_the_boundary_accepts():
    m = _mat(4)
    for src in (SparseRowArray.from_scipy(m), SparseRowArray.from_scipy(m, large=True),
                _rust_chunk(m), m, pa.chunked_array([_rust_chunk(m)])):
        off, idx, val, shape = csr_arrays(src)
        assert shape == m.shape and np.array_equal(off, m.indptr)
        assert np.array_equal(idx, m.indices) and np.array_equal(val, m.data)
    # zero-copy: the NumPy views alias the Arrow buffers
    a = SparseRowArray.from_scipy(m)
    _off, idx, _val, _ = csr_arrays(a)
    assert idx.ctypes.data == a.indices.buffers()[1].address + a.indices.offset * 4
    # a slice rebases its offsets
    off, idx, val, shape = csr_arrays(a.slice(10, 5))
    sub = m[10:15]
    assert shape == (5, 30) and np.array_equal(off, sub.indptr) and np.array_equal(val, sub.data)
    # structure only
    off, idx, val, shape = csr_arrays(SparseRowArray.from_scipy(m, values=False))
    assert val is None and np.array_equal(idx, m.indices)


def test_legacy_layout_and_type_errors():
    "csr.rs:161-193 / matrix.py:364-385: wrong Arrow types are TypeErrors"
    m = _mat(5)
    legacy = pa.ListArray.from_arrays(
        pa.array(m.indptr.astype(np.int32)),
        pa.StructArray.from_arrays([pa.array(m.indices, pa.int32()), pa.array(m.data)],
                                   names=["index", "value"]))
    with pytest.raises(TypeError):
        csr_arrays(legacy)  # no dimension anywhere
    a = SparseRowArray.from_array(legacy, dimension=30)
    assert a.shape == (50, 30) and np.array_equal(a.values.to_numpy(), m.data)
    with pytest.raises(ValueError):
        SparseRowArray.from_array(SparseRowArray.from_scipy(m), dimension=31)
    with pytest.raises(TypeError):
        csr_arrays(pa.array([1.0, 2.0]))
    with pytest.raises(TypeError):
        csr_arrays(np.zeros(3))
    bad = pa.ListArray.from_arrays(
        pa.array(m.indptr.astype(np.int32)),
        pa.StructArray.from_arrays([pa.array(m.indices, pa.int32()), pa.array(m.data)],
                                   names=["col", "value"]))
    with pytest.raises(TypeError):
        SparseRowArray.from_array(bad, dimension=30)


def test_int64_offsets_when_large_requested():
    m = _mat(6)
    a = SparseRowArray.from_arrays(m.indptr.astype(np.int64), m.indices, m.data, shape=m.shape)
    assert pa.types.is_large_list(a.type.storage_type)
    off, _i, _v, _s = csr_arrays(a)
    assert off.dtype == np.int64
