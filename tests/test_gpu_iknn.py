"""GPU parity: item-item similarity build vs the CPU oracle -- BIT-EXACT."""
import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu


def _build_gpu(D, gpu, ui, iu, min_sim, is64=False):
    dt = np.int64 if is64 else np.int32
    dui = D.DeviceCSR.from_arrays(ui.indptr.astype(dt), ui.indices, ui.data, ui.shape, gpu)
    diu = D.DeviceCSR.from_arrays(iu.indptr.astype(dt), iu.indices, iu.data, iu.shape, gpu)
    out = D.iknn_build(dui, diu, min_sim)
    return (out.indptr.cpu().numpy(), out.indices.cpu().numpy(), out.values.cpu().numpy())


def _assert_same(got, want: sps.csr_array):
    ptr, idx, val = got
    assert ptr.dtype == np.int64  # LargeList offsets (src/lenskit/knn/item.py:176)
    assert np.array_equal(ptr, want.indptr)
    assert np.array_equal(idx, want.indices)
    # bit-exact values: same products, same order, no FMA contraction
    assert np.array_equal(val.view(np.uint32), want.data.astype(np.float32).view(np.uint32))


@pytest.mark.parametrize("explicit", [True, False])
@pytest.mark.parametrize("is64", [False, True])
def test_build_ml_small_bit_exact(gpu, oracle, ml_small, explicit, is64):
    from lkpy_amd import _device as D

    rmat = ml_small["rmat"]
    if not explicit:
        rmat = sps.coo_array((np.ones(rmat.nnz, np.float32), (rmat.row, rmat.col)), rmat.shape)
    ui, iu, _means, _ = oracle.iknn_prepare(rmat, explicit)
    want = oracle.iknn_build(ui, iu, 1.0e-6, None)
    got = _build_gpu(D, gpu, ui, iu, 1.0e-6, is64)
    _assert_same(got, want)
    ptr, idx, val = got
    assert val.min() > 0 and val.max() <= 1 + 1e-6  # tests/models/test_knn_item_item.py:134-148
    # no self-similarity (item_train.rs:120-122); rows sorted by column (item_train.rs:149)
    rows = np.repeat(np.arange(len(ptr) - 1), np.diff(ptr))
    assert not np.any(rows == idx)
    same_row = rows[1:] == rows[:-1]
    assert np.all(np.diff(idx)[same_row] > 0)


def test_build_windows_and_edges(gpu, oracle, rng):
    """More items than one LDS window (several tasks per row), empty users and items,
    negative values, a threshold that cuts through the distribution."""
    from lkpy_amd import _device as D
    from lkpy_amd import synth

    mat = synth.ml25m_like(seed=7, scale=0.15)  # ~9.3k items > W = 8192 -> P = 2
    assert mat.shape[1] > 8192
    ui, iu, _m, _ = oracle.iknn_prepare(sps.coo_array(mat), True)
    for min_sim in (1.0e-6, 0.05):
        want = oracle.iknn_build(ui, iu, min_sim, None)
        got = _build_gpu(D, gpu, ui, iu, min_sim)
        _assert_same(got, want)


def test_build_two_pass_fallback(gpu, oracle, ml_small, monkeypatch):
    """Without room for the n_items^2 staging area the build runs as two full passes
    (count, fill); same bits."""
    from lkpy_amd import _device as D

    monkeypatch.setenv("LK_IKNN_STAGE_GB", "0")
    ui, iu, _means, _ = oracle.iknn_prepare(ml_small["rmat"], True)
    want = oracle.iknn_build(ui, iu, 1.0e-6, None)
    _assert_same(_build_gpu(D, gpu, ui, iu, 1.0e-6), want)


def test_build_wide_addressing(gpu, oracle, ml_small, monkeypatch):
    "The addressing path for nnz >= 2^29 (64-bit chunk addresses), forced on a small input."
    from lkpy_amd import _device as D

    monkeypatch.setenv("LK_IKNN_ADDR64", "1")
    ui, iu, _means, _ = oracle.iknn_prepare(ml_small["rmat"], True)
    want = oracle.iknn_build(ui, iu, 1.0e-6, None)
    _assert_same(_build_gpu(D, gpu, ui, iu, 1.0e-6), want)


@pytest.mark.parametrize("save_nbrs", [None, 20])
def test_build_row_shards(gpu, oracle, ml_small, save_nbrs):
    """Multi-GPU sharding of the build: every rank builds a block of output rows with no
    collective; the blocks stacked are bit-identical to the full build."""
    from lkpy_amd import _device as D

    ui, iu, _means, _ = oracle.iknn_prepare(ml_small["rmat"], True)
    want = oracle.iknn_build(ui, iu, 1.0e-6, save_nbrs)
    dui, diu = D.DeviceCSR.from_scipy(ui, gpu), D.DeviceCSR.from_scipy(iu, gpu)
    n = ui.shape[1]
    cuts = [0, n // 3, n // 3, (2 * n) // 3 + 5, n]  # uneven blocks incl. an empty one
    ptrs, idxs, vals = [np.zeros(1, np.int64)], [], []
    base = 0
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        part = D.iknn_build(dui, diu, 1.0e-6, save_nbrs, rows=(lo, hi))
        assert part.shape == (hi - lo, n)
        p = part.indptr.cpu().numpy()
        assert p[0] == 0 and len(p) == hi - lo + 1
        ptrs.append(p[1:] + base)
        base += int(p[-1])
        idxs.append(part.indices.cpu().numpy())
        vals.append(part.values.cpu().numpy())
    _assert_same((np.concatenate(ptrs), np.concatenate(idxs), np.concatenate(vals)), want)


def test_build_toy_closed_form(gpu, oracle):
    """The reference's 14-rating toy set: sim(6,7) equals the hand-computed centred
    cosine (tests/models/test_knn_item_item.py:106-162)."""
    from lkpy_amd import _device as D

    recs = [(1, 6, 4.0), (2, 6, 2.0), (1, 7, 3.0), (2, 7, 2.0), (3, 7, 5.0), (4, 7, 2.0),
            (1, 8, 3.0), (2, 8, 4.0), (3, 8, 3.0), (4, 8, 2.0), (5, 8, 3.0), (6, 8, 2.0),
            (1, 9, 3.0), (3, 9, 4.0)]  # fmt: skip
    u = np.array([r[0] - 1 for r in recs])
    i = np.array([r[1] - 6 for r in recs])
    v = np.array([r[2] for r in recs], np.float32)
    rmat = sps.coo_array((v, (u, i)), shape=(6, 4))
    ui, iu, means, _ = oracle.iknn_prepare(rmat, True)
    ptr, idx, val = _build_gpu(D, gpu, ui, iu, 1.0e-6)
    S = sps.csr_array((val, idx, ptr), shape=(4, 4)).toarray()
    six = np.array([4.0, 2.0]) - 3.0
    seven = np.array([3.0, 2.0, 5.0, 2.0]) - 3.0
    num = six[0] * seven[0] + six[1] * seven[1]
    denom = np.linalg.norm(six) * np.linalg.norm(seven)
    assert S[0, 1] == pytest.approx(num / denom, rel=1e-5)
    assert S[0, 1] == S[1, 0]
    assert np.all(val > 0)


def test_build_empty(gpu):
    from lkpy_amd import _device as D

    ui = sps.csr_array((3, 5), dtype=np.float32)
    iu = sps.csr_array((5, 3), dtype=np.float32)
    ptr, idx, val = _build_gpu(D, gpu, ui, iu, 1e-6)
    assert np.array_equal(ptr, np.zeros(6, np.int64)) and len(idx) == 0 and len(val) == 0


@pytest.mark.parametrize("save_nbrs", [1, 20, 100, 500])
@pytest.mark.parametrize("explicit", [True, False])
def test_save_nbrs_bit_exact(gpu, oracle, ml_small, save_nbrs, explicit):
    """Per-row truncation (item_train.rs:139-151) incl. the first-encounter tie order: the
    implicit model has MANY exact ties (equal co-rating patterns) across the cut."""
    from lkpy_amd import _device as D

    rmat = ml_small["rmat"]
    if not explicit:
        rmat = sps.coo_array((np.ones(rmat.nnz, np.float32), (rmat.row, rmat.col)), rmat.shape)
    ui, iu, _m, _ = oracle.iknn_prepare(rmat, explicit)
    want = oracle.iknn_build(ui, iu, 1.0e-6, save_nbrs)
    dui, diu = D.DeviceCSR.from_scipy(ui, gpu), D.DeviceCSR.from_scipy(iu, gpu)
    out = D.iknn_build(dui, diu, 1.0e-6, save_nbrs)
    got = (out.indptr.cpu().numpy(), out.indices.cpu().numpy(), out.values.cpu().numpy())
    _assert_same(got, want)
    assert np.all(np.diff(got[0]) <= save_nbrs)
    # the test really exercises ties at the cut
    full = oracle.iknn_build(ui, iu, 1.0e-6, None)
    ties = 0
    for i in np.flatnonzero(np.diff(full.indptr) > save_nbrs)[:400]:
        v = np.sort(full.data[full.indptr[i] : full.indptr[i + 1]])[::-1]
        ties += int(v[save_nbrs - 1] == v[save_nbrs])
    if not explicit and save_nbrs >= 20:
        assert ties > 0


def test_save_nbrs_rejected_by_raw_build(gpu, oracle, ml_small):
    "The raw build entry points take save_nbrs <= 0 only and say so loudly."
    import ctypes

    from lkpy_amd import _device as D
    from lkpy_amd import _native

    lib = _native.load()
    assert lib.lk_iknn_truncate_count(None, None, None, None, 0, None, 5, 0, 5, 0, None, None,
                                      ctypes.byref(ctypes.c_int64(0)), None) == _native.LK_E_INVALID


def test_symmetric_build_equals_full_build(gpu, rng, monkeypatch):
    """
    The symmetric build (windows on / right of the diagonal accumulated, the others mirrored
    from their transposes) and the full build give the same CSR, bit for bit: multi-window
    matrix with a partial last window, empty items, heavy items.
    """
    from lkpy_amd import _device as D

    monkeypatch.setenv("LK_IKNN_W", "1024")
    n_users, n_items = 3000, 5 * 1024 + 333
    lens = np.clip(rng.geometric(1 / 25.0, n_users), 1, 400)
    rows = np.repeat(np.arange(n_users), lens)
    pop = rng.zipf(1.3, len(rows)) % n_items
    cols = (pop * 7919) % n_items
    m = sps.csr_array((np.ones(len(rows), np.float32), (rows, cols)), shape=(n_users, n_items))
    m.sum_duplicates()
    m.data = rng.random(m.nnz).astype(np.float32) + 0.1
    m.sort_indices()
    iu = sps.csr_array(m.T)
    iu.sort_indices()
    dui, diu = D.DeviceCSR.from_scipy(m, gpu), D.DeviceCSR.from_scipy(iu, gpu)
    outs = {}
    for sym in ("1", "0"):
        monkeypatch.setenv("LK_IKNN_SYMMETRIC", sym)
        o = D.iknn_build(dui, diu, 1.0e-6, None)
        outs[sym] = (o.indptr.cpu().numpy(), o.indices.cpu().numpy(), o.values.cpu().numpy())
    assert outs["1"][0][-1] > 100000
    for a, b in zip(outs["1"], outs["0"]):
        assert np.array_equal(a.view(np.uint8) if a.dtype == np.float32 else a,
                              b.view(np.uint8) if b.dtype == np.float32 else b)
