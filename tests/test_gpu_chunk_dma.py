"""
k = 256: the chunk kernel that stages the gathered rows through LDS (als_blk_chunk_dma_kernel,
csrc/als_blk.hip) against the register-ring chunk kernel it replaces -- same MFMA sequence per
entry group, so the half-epoch must come out BIT-identical (`LK_BLK_CHUNK_DMA=0` selects the old
kernel), for chunk lengths that end everywhere in a 16-row stage / a 64-entry batch, and against
the oracle (src/accel/als/implicit.rs:87-125, explicit.rs:33-119).
"""
import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu

LENS = [2049, 2050, 2063, 2064, 2065, 2111, 2112, 2113, 3000, 4096, 4097, 5121, 9999, 20000,
        1, 7, 100, 2048, 0]


def _csr(rng, n_cols):
    indptr = np.zeros(len(LENS) + 1, np.int64)
    np.cumsum(LENS, out=indptr[1:])
    indices = np.concatenate(
        [np.sort(rng.choice(n_cols, n, replace=False)) for n in LENS]).astype(np.int32)
    values = rng.integers(1, 6, indptr[-1]).astype(np.float32)
    return sps.csr_array((values, indices, indptr), shape=(len(LENS), n_cols))


@pytest.mark.parametrize("k", [200, 256])
def test_implicit_bit_identical_and_vs_oracle(gpu, oracle, rng, monkeypatch, k):
    from lkpy_amd import _device as D
    from lkpy_amd import _native

    n_cols = 30000
    mat = _csr(rng, n_cols)
    other = (rng.standard_normal((n_cols, k)) * 0.05).astype(np.float32)
    this = np.zeros((mat.shape[0], k), np.float32)
    otor = oracle.implicit_otor(other, 0.1)
    want = this.copy()
    oracle.als_half_epoch(mat, want, other, otor)
    csr = D.DeviceCSR.from_arrays(mat.indptr.astype(np.int32), mat.indices, mat.data, mat.shape, gpu)
    d_other = D.to_device_padded(other, gpu)
    d_otor = D.Gramian(k, gpu)(d_other, 0.1)
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("LK_BLK_CHUNK_DMA", mode)
        plan = D.ALSPlan(csr, k, _native.SOLVER_CHOLESKY)
        d_this = D.to_device_padded(this, gpu)
        plan.half_epoch(d_this, d_other, d_otor)
        plan.check_status()
        got[mode] = D.to_host_unpadded(d_this, k)
    assert np.array_equal(got["1"], got["0"])
    err = np.linalg.norm(got["1"] - want, axis=1)
    assert np.all(err <= 5e-4 * np.maximum(np.linalg.norm(want, axis=1), 1e-3))


def test_explicit_bit_identical(gpu, rng, monkeypatch):
    from lkpy_amd import _device as D
    from lkpy_amd import _native

    k, n_cols = 256, 30000
    mat = _csr(rng, n_cols)
    mat.data[:] = rng.standard_normal(mat.data.shape[0]).astype(np.float32)
    other = (rng.standard_normal((n_cols, k)) * 0.05).astype(np.float32)
    csr = D.DeviceCSR.from_arrays(mat.indptr.astype(np.int32), mat.indices, mat.data, mat.shape, gpu)
    d_other = D.to_device_padded(other, gpu)
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("LK_BLK_CHUNK_DMA", mode)
        plan = D.ALSPlan(csr, k, _native.SOLVER_CHOLESKY)
        d_this = D.to_device_padded(np.zeros((mat.shape[0], k), np.float32), gpu)
        plan.half_epoch_explicit(d_this, d_other, 0.1)
        plan.check_status()
        got[mode] = D.to_host_unpadded(d_this, k)
    assert np.isfinite(got["1"]).all()
    assert np.array_equal(got["1"], got["0"])
