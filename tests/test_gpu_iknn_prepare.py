"""GPU parity: item-kNN rating normalisation on the device vs the reference's SciPy calls
(restated in oracle.iknn_prepare) -- BIT-EXACT values, identical structure."""
import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu


def _same(d, want: sps.csr_array):
    assert np.array_equal(d.indptr.cpu().numpy(), want.indptr)
    assert np.array_equal(d.indices.cpu().numpy(), want.indices)
    got = d.values.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.data.astype(np.float32).view(np.uint32))


@pytest.mark.parametrize("explicit", [True, False])
def test_prepare_ml_small_bit_exact(gpu, oracle, ml_small, explicit):
    from lkpy_amd import _device as D

    rmat = ml_small["rmat"]
    if not explicit:  # the interaction matrix of an implicit-feedback model holds ones
        rmat = sps.coo_array((np.ones(rmat.nnz, np.float32), (rmat.row, rmat.col)), rmat.shape)
    ui, iu, means, zero = oracle.iknn_prepare(rmat, explicit)
    dui, diu, dmeans, dzero = D.iknn_prepare(rmat, explicit, gpu)
    _same(dui, ui)
    _same(diu, iu)
    assert dzero == zero
    if explicit:
        assert np.array_equal(dmeans.view(np.uint32),
                              np.asarray(means, np.float32).ravel().view(np.uint32))
    else:
        assert dmeans is None
    # and the build from it equals the build from the host-prepared matrices
    want = oracle.iknn_build(ui, iu, 1.0e-6, None)
    out = D.iknn_build(dui, diu, 1.0e-6)
    assert np.array_equal(out.indptr.cpu().numpy(), want.indptr)
    assert np.array_equal(out.values.cpu().numpy().view(np.uint32),
                          want.data.astype(np.float32).view(np.uint32))


def test_prepare_synthetic_with_empty_items_and_heavy_tail(gpu, oracle):
    "Items nobody rated (norm clamp), items with > 64 ratings (several chunks per wave)."
    from lkpy_amd import _device as D
    from lkpy_amd import synth

    mat = synth.ml25m_like(seed=11, scale=0.1)
    assert (np.bincount(mat.indices, minlength=mat.shape[1]) == 0).any()
    ui, iu, means, _ = oracle.iknn_prepare(sps.coo_array(mat), True)
    dui, diu, dmeans, _ = D.iknn_prepare(mat, True, gpu)
    _same(dui, ui)
    _same(diu, iu)
    assert np.array_equal(dmeans.view(np.uint32),
                          np.asarray(means, np.float32).ravel().view(np.uint32))


def test_prepare_constant_ratings_flag(gpu):
    "All ratings equal: centred values are all zero -> the reference warns (item.py:211-216)."
    from lkpy_amd import _device as D

    m = sps.random(50, 30, density=0.3, format="csr", dtype=np.float32, random_state=4)
    m.data[:] = 3.0
    _ui, _iu, _means, zero = D.iknn_prepare(m, True, gpu)
    assert zero
