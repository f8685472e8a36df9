"""GPU parity: stable CSR transpose vs the oracle's restatement of transpose.rs -- BIT-EXACT."""
import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("is64", [False, True])
@pytest.mark.parametrize("shape", [(1, 1), (300, 7), (7, 300), (5000, 70001)])
def test_transpose_bit_exact(gpu, oracle, rng, shape, is64):
    from lkpy_amd import _device as D

    n_rows, n_cols = shape
    m = sps.random(n_rows, n_cols, density=min(0.3, 2000.0 / (n_rows * n_cols) * 50),
                   format="csr", dtype=np.float32, random_state=int(rng.integers(1 << 30)))
    m.sort_indices()
    m.data = rng.standard_normal(m.nnz).astype(np.float32)
    dt = np.int64 if is64 else np.int32
    csr = D.DeviceCSR.from_arrays(m.indptr.astype(dt), m.indices, m.data, m.shape, gpu)
    t = D.csr_transpose(csr)
    ptr, idx, perm = oracle.transpose_csr(m.indptr.astype(dt), m.indices, n_cols)
    assert t.shape == (n_cols, n_rows)
    assert t.indptr.cpu().numpy().dtype == dt
    assert np.array_equal(t.indptr.cpu().numpy(), ptr)
    assert np.array_equal(t.indices.cpu().numpy(), idx)
    assert np.array_equal(t.perm.cpu().numpy(), perm)
    assert np.array_equal(t.values.cpu().numpy().view(np.uint32), m.data[perm].view(np.uint32))
    # transposing twice is the identity (columns of a row were sorted)
    tt = D.csr_transpose(t)
    assert np.array_equal(tt.indptr.cpu().numpy(), m.indptr.astype(dt))
    assert np.array_equal(tt.indices.cpu().numpy(), m.indices)
    assert np.array_equal(tt.values.cpu().numpy(), m.data)


def test_transpose_empty_and_structure_only(gpu, oracle):
    from lkpy_amd import _device as D
    from lkpy_amd.data import SparseRowArray

    e = sps.csr_array((4, 6), dtype=np.float32)
    t = D.csr_transpose(D.DeviceCSR.from_scipy(e, gpu))
    assert np.array_equal(t.indptr.cpu().numpy(), np.zeros(7, e.indptr.dtype)) and t.nnz == 0
    # the host data class goes through the same kernel (matrix.py:512-530)
    m = sps.random(40, 25, density=0.2, format="csr", dtype=np.float32, random_state=1)
    sra = SparseRowArray.from_scipy(m)
    tr = sra.transpose()
    want = sps.csr_array(m.T)
    want.sort_indices()
    assert tr.shape == (25, 40)
    assert np.array_equal(tr.offsets.to_numpy(), want.indptr)
    assert np.array_equal(tr.indices.to_numpy(), want.indices)
    assert np.array_equal(tr.values.to_numpy(), want.data)
    s_only = SparseRowArray.from_scipy(m, values=False).transpose()
    assert s_only.values is None and np.array_equal(s_only.indices.to_numpy(), want.indices)


@pytest.mark.parametrize("dtype,n", [("int32", 50_000_003), ("float32", 16_777_216 + 5), ("int64", 1000)])
def test_to_host_staged_download(gpu, dtype, n):
    "D.to_host: the pinned-ring / thread-team download returns exactly what .cpu() does"
    import torch

    from lkpy_amd import _device as D

    g = torch.Generator(device=gpu).manual_seed(3)
    if dtype == "float32":
        t = torch.randn(n, device=gpu, generator=g)
    else:
        t = torch.randint(-2**31 + 1, 2**31 - 1, (n,), device=gpu, generator=g,
                          dtype=getattr(torch, dtype))
    got = D.to_host(t, threads=6)
    assert got.dtype == np.dtype(dtype) and got.shape == (n,)
    assert np.array_equal(got, t.cpu().numpy())
    # a second transfer reuses the ring; a 2-D view keeps its shape
    t2 = t[: (n // 7) * 7].reshape(-1, 7)
    assert np.array_equal(D.to_host(t2), t2.cpu().numpy())


@pytest.mark.parametrize("n_rows,n_cols,nnz", [(20000, 300, 3_000_000), (3000, 70001, 2_500_000),
                                                (10, 5, 37), (1, 100000, 4097), (4096, 4096, 4096)])
def test_transpose_hand_written_sort_sizes(gpu, oracle, rng, n_rows, n_cols, nnz):
    """Round 6: the transpose's stable radix sort is this repository's own (csrc/radix_sort.h: 8-bit
    digits, 4096-key tiles; rounds 1-5 called rocPRIM).  Sizes that end inside a tile, span
    hundreds of tiles, take one / two / three digit passes; heavy duplicate columns (stability is
    what keeps an output row's entries in source-row order): offsets, indices and the permutation
    equal to the oracle's counting-sort transpose, entry for entry."""
    from lkpy_amd import _device as D

    rows = np.sort(rng.integers(0, n_rows, nnz))
    cols = rng.integers(0, n_cols, nnz)
    m = sps.csr_array((np.ones(nnz, np.float32), (rows, cols)), shape=(n_rows, n_cols))
    m.sum_duplicates()
    m.sort_indices()
    m.data = rng.standard_normal(m.nnz).astype(np.float32)
    csr = D.DeviceCSR.from_arrays(m.indptr, m.indices, m.data, m.shape, gpu)
    t = D.csr_transpose(csr)
    ptr, idx, perm = oracle.transpose_csr(m.indptr, m.indices, n_cols)
    assert np.array_equal(t.indptr.cpu().numpy(), ptr)
    assert np.array_equal(t.indices.cpu().numpy(), idx)
    assert np.array_equal(t.perm.cpu().numpy(), perm)
