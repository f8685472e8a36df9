"""GPU parity: explicit-feedback (biased-MF) ALS half-epoch vs the CPU oracle."""
import numpy as np
import pytest
import scipy.sparse as sps

from test_gpu_als import RTOL, _random_csr, _rel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k", [10, 25, 64, 128, 200])
@pytest.mark.parametrize("is64", [False, True])
def test_explicit_half_epoch_random(gpu, oracle, rng, k, is64):
    """explicit.rs:80-119 on the kernel: A = M^T M + reg n I, rhs M^T r; short rows, rows that
    go through the chunk kernel (> 2048 entries), empty rows (zeros, no delta)."""
    from lkpy_amd import _device as D
    from lkpy_amd import _native

    n_rows, n_cols = 3000, 5000
    mat = _random_csr(rng, n_rows, n_cols, 30, long_rows=(2049, 4100, 5000, 2048))
    mat.data = rng.standard_normal(mat.nnz).astype(np.float32)  # bias-normalised ratings
    other = oracle.als_explicit_initial_params(rng, n_cols, k)
    this = oracle.als_explicit_initial_params(rng, n_rows, k)
    reg = 0.1

    want = this.copy()
    want_frob = oracle.als_explicit_half_epoch(mat, want, other, reg)
    exact = oracle.als_explicit_half_epoch_f64(mat, other, reg)

    indptr = mat.indptr.astype(np.int64 if is64 else np.int32)
    csr = D.DeviceCSR.from_arrays(indptr, mat.indices, mat.data, mat.shape, gpu)
    plan = D.ALSPlan(csr, k, _native.SOLVER_CHOLESKY)
    d_this = D.to_device_padded(this, gpu)
    d_other = D.to_device_padded(other, gpu)
    frob = plan.half_epoch_explicit(d_this, d_other, reg)
    plan.check_status()
    got = D.to_host_unpadded(d_this, k)

    empty = np.diff(mat.indptr) == 0
    assert empty.any() and np.all(got[empty] == 0.0)  # explicit.rs:91-94
    assert _rel(got, want) < RTOL
    # at least as close to the float64 answer as the reference arithmetic
    assert _rel(got, exact) <= 2 * _rel(want, exact) + 1e-6
    rn = np.linalg.norm(want, axis=1)
    err = np.linalg.norm(got - want, axis=1)
    assert np.all(err <= 5 * RTOL * np.maximum(rn, 1e-3))
    assert abs(float(frob.item()) - want_frob) <= 1e-4 * want_frob
    if d_this.shape[1] > k:
        assert float(d_this[:, k:].abs().max().item()) == 0.0
    # bit-reproducible
    d_this2 = D.to_device_padded(this, gpu)
    plan.half_epoch_explicit(d_this2, d_other, reg)
    plan.check_status()
    assert np.array_equal(D.to_host_unpadded(d_this2, k), got)


def test_explicit_refuses_cg_plan(gpu, rng):
    "Only the exact solver exists for the explicit model; a CG plan says so loudly."
    from lkpy_amd import _device as D
    from lkpy_amd import _native

    mat = _random_csr(rng, 50, 80, 5)
    plan = D.ALSPlan(D.DeviceCSR.from_scipy(mat, gpu), 64, _native.SOLVER_CG)
    z = D.to_device_padded(np.zeros((50, 64), np.float32), gpu)
    o = D.to_device_padded(np.zeros((80, 64), np.float32), gpu)
    with pytest.raises(ValueError, match="explicit"):
        plan.half_epoch_explicit(z, o, 0.1)
