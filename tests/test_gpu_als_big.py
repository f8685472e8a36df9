"""
Embedding sizes above 256 (csrc/als_big.hip): the blocked Cholesky of als_blk.hip on 16 x 16 tiles
kept in an HBM scratch -- `POSV::solve` (src/accel/als/solve.rs:65-107) takes any k, so does this
path up to 1024 (padded to a multiple of 64).  Against the oracle's sposv restatement
(src/accel/als/implicit.rs:87-125; explicit.rs:80-119), tolerance 1e-4 relative (north star).
"""
import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu

RTOL = 1e-4


def _rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _csr(rng, n_rows, n_cols, lens_head, mean_len=40, values=40.0):
    lens = np.clip(rng.geometric(1.0 / mean_len, n_rows), 1, n_cols)
    lens[: len(lens_head)] = lens_head
    indptr = np.zeros(n_rows + 1, np.int64)
    np.cumsum(lens, out=indptr[1:])
    indices = np.concatenate(
        [np.sort(rng.choice(n_cols, ln, replace=False)) for ln in lens]).astype(np.int32)
    data = np.full(indptr[-1], values, np.float32) if np.isscalar(values) else \
        values(indptr[-1]).astype(np.float32)
    return sps.csr_array((data, indices, indptr), shape=(n_rows, n_cols))


@pytest.mark.parametrize("k,n", [(300, 5), (300, 3001), (512, 70_001), (1000, 257)])
def test_gramian_above_256(gpu, rng, k, n):
    from lkpy_amd import _device as D

    m = (rng.standard_normal((n, k)) * 0.1).astype(np.float32)
    got = D.Gramian(k, gpu)(D.to_device_padded(m, gpu), 0.25).cpu().numpy()
    want = m.astype(np.float64).T @ m.astype(np.float64) + 0.25 * np.eye(k)
    assert got.shape == (k, k)
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max() + 1e-7 * np.sqrt(n)
    assert np.array_equal(got, got.T)


@pytest.mark.parametrize("k,is64", [(257, False), (320, False), (384, True), (512, False)])
def test_half_epoch_above_256(gpu, oracle, rng, k, is64):
    """Implicit half-epoch at k = 257 (padded to 320), 320, 384, 512: empty rows, rows shorter than
    one MFMA step, a 3 000-entry row, both offset widths; bit-reproducible; pads stay zero."""
    from lkpy_amd import _device as D
    from lkpy_amd import _native

    n_rows, n_cols = 160, 3500
    mat = _csr(rng, n_rows, n_cols, [3000, 700, 0, 1, 2, 3, 5, 17, 64, 65, 0, 300])
    other = (rng.standard_normal((n_cols, k)) * 0.1).astype(np.float32)
    this = (rng.standard_normal((n_rows, k)) * 0.1).astype(np.float32)
    otor = oracle.implicit_otor(other, 0.1)
    want = this.copy()
    want_frob = oracle.als_half_epoch(mat, want, other, otor)
    exact, _ = oracle.als_referee_f64(mat, other, 0.1, with_cond=False)

    dt = np.int64 if is64 else np.int32
    csr = D.DeviceCSR.from_arrays(mat.indptr.astype(dt), mat.indices, mat.data, mat.shape, gpu)
    plan = D.ALSPlan(csr, k, _native.SOLVER_AUTO)
    assert plan.solver == _native.SOLVER_CHOLESKY and plan.kp % 64 == 0 and not plan.use_wb
    d_this = D.to_device_padded(this, gpu)
    d_other = D.to_device_padded(other, gpu)
    d_otor = D.Gramian(k, gpu)(d_other, 0.1)
    frob = plan.half_epoch(d_this, d_other, d_otor)
    plan.check_status()
    got = D.to_host_unpadded(d_this, k)
    empty = np.diff(mat.indptr) == 0
    assert empty.any() and np.all(got[empty] == 0.0)
    rn = np.linalg.norm(want, axis=1)
    err = np.linalg.norm(got - want, axis=1)
    print(f"\nk={k}: rel {_rel(got, want):.2e}, worst row {float((err / np.maximum(rn, 1e-30))[~empty].max()):.2e}; "
          f"vs float64: gpu {_rel(got, exact):.2e} oracle {_rel(want, exact):.2e}")
    assert _rel(got, want) < RTOL
    assert np.all(err <= 5 * RTOL * np.maximum(rn, 1e-3))
    assert _rel(got, exact) <= 2 * _rel(want, exact) + 1e-6
    assert abs(float(frob.item()) - want_frob) <= 1e-4 * want_frob
    if d_this.shape[1] > k:
        assert float(d_this[:, k:].abs().max().item()) == 0.0
    d2 = D.to_device_padded(this, gpu)
    plan.half_epoch(d2, d_other, d_otor)
    plan.check_status()
    assert np.array_equal(D.to_host_unpadded(d2, k), got)


def test_explicit_half_epoch_above_256(gpu, oracle, rng):
    "explicit.rs:80-119 at k = 320: A = M^T M + reg n I, y = M^T r"
    from lkpy_amd import _device as D
    from lkpy_amd import _native

    k, n_rows, n_cols = 320, 90, 2000
    mat = _csr(rng, n_rows, n_cols, [900, 0, 1, 400], mean_len=30,
               values=lambda n: rng.normal(0.0, 1.0, n))
    other = (rng.standard_normal((n_cols, k)) * 0.1).astype(np.float32)
    this = np.zeros((n_rows, k), np.float32)
    want = this.copy()
    oracle.als_explicit_half_epoch(mat, want, other, 0.05)
    csr = D.DeviceCSR.from_arrays(mat.indptr.astype(np.int32), mat.indices, mat.data, mat.shape,
                                  gpu)
    plan = D.ALSPlan(csr, k, _native.SOLVER_CHOLESKY)
    d_this = D.to_device_padded(this, gpu)
    plan.half_epoch_explicit(d_this, D.to_device_padded(other, gpu), 0.05)
    plan.check_status()
    got = D.to_host_unpadded(d_this, k)
    nz = np.diff(mat.indptr) > 0
    err = np.linalg.norm(got[nz] - want[nz], axis=1) / np.linalg.norm(want[nz], axis=1)
    assert err.max() < 5 * RTOL and _rel(got, want) < RTOL, (err.max(), _rel(got, want))
    assert not got[~nz].any()


def test_not_spd_and_limits_above_256(gpu):
    import torch

    from lkpy_amd import _device as D
    from lkpy_amd import _native
    from lkpy_amd.als import ImplicitMFScorer

    k = 320
    mat = sps.csr_array((np.array([1.0], np.float32), np.array([0], np.int32),
                         np.array([0, 1], np.int64)), shape=(1, 2))
    plan = D.ALSPlan(D.DeviceCSR.from_scipy(mat, gpu), k, _native.SOLVER_CHOLESKY)
    this = torch.zeros((1, k), device=gpu)
    plan.half_epoch(this, torch.ones((2, k), device=gpu), -torch.eye(k, device=gpu) * 100.0)
    with pytest.raises(RuntimeError, match="ALS solve error"):
        plan.check_status()
    with pytest.raises(Exception):  # the CG option stops at 256
        D.ALSPlan(D.DeviceCSR.from_scipy(mat, gpu), k, _native.SOLVER_CG)
    with pytest.raises(ValueError):
        ImplicitMFScorer(embedding_size=1025)
    assert ImplicitMFScorer(embedding_size=1024).config.embedding_size == 1024


def test_component_trains_and_recommends_at_k_300(gpu, oracle):
    """``ImplicitMFScorer(embedding_size=300)`` end to end: training epochs against the oracle from
    the same draws, fold-in + dense scoring + top-N (panel path at padded k = 320)."""
    from lkpy_amd.als import ImplicitMFScorer
    from lkpy_amd.data import Dataset, RecQuery
    from lkpy_amd.training import TrainingOptions

    rng = np.random.default_rng(4)
    n_users, n_items, k = 400, 500, 300
    users = rng.integers(0, n_users, 12000)
    items = rng.integers(0, n_items, 12000)
    ds = Dataset.from_arrays(users, items, np.ones(len(users), np.float32))
    sc = ImplicitMFScorer(embedding_size=k, epochs=2)
    sc.train(ds, TrainingOptions(rng=7))
    assert sc.item_embeddings.shape == (ds.item_count, k) and np.isfinite(sc.item_embeddings).all()
    # the same two epochs on the oracle from the same draws
    ui = ds.interactions().matrix().scipy(layout="csr")
    ui = sps.csr_array((np.full(ui.nnz, 40.0, np.float32), ui.indices, ui.indptr), shape=ui.shape)
    ui.sum_duplicates()
    iu = sps.csr_array(ui.T)
    iu.sort_indices()
    g = np.random.default_rng(7)
    Q = oracle.als_initial_params(g, ds.item_count, k)
    P = oracle.als_initial_params(g, ds.user_count, k)
    for _ in range(2):
        oracle.als_half_epoch(ui, P, Q, oracle.implicit_otor(Q, 0.1))
        oracle.als_half_epoch(iu, Q, P, oracle.implicit_otor(P, 0.1))
    # (two epochs from the tiny init are ill-conditioned: a loose matrix-level bound)
    assert _rel(sc.item_embeddings, Q) < 2e-2 and _rel(sc.user_embeddings, P) < 2e-2
    # recommend: top-10 for a few users == argsort of the scorer's own dense scores
    qs = [RecQuery(user_id=int(u), user_items=ds.user_row(int(u))) for u in ds.users._ids[:5]]
    idx, scv = sc.recommend_batch(qs, 10)
    for q, row_i, row_s in zip(qs, idx, scv):
        from lkpy_amd.data import ItemList

        full = sc(q, ItemList(item_nums=np.arange(ds.item_count), vocabulary=sc.items)).scores()
        own = q.user_items.numbers(vocabulary=sc.items)
        full = np.asarray(full, np.float32).copy()
        full[own] = -np.inf
        top = np.argsort(-full, kind="stable")[:10]
        assert np.array_equal(np.sort(full[top])[::-1].view(np.uint32), row_s.view(np.uint32))
        assert set(row_i.tolist()) == set(top.tolist()) or \
            np.array_equal(full[row_i].view(np.uint32), row_s.view(np.uint32))


def test_long_row_above_256_follows_the_reference_order(gpu, oracle):
    """a 120 000-entry row at k = 320 whose gathered factor rows repeat (the case in
    which the reference's sequential float32 sums drift systematically, tests/
    test_gpu_als_rhs_order.py).  The k > 256 kernels now sum every row in the reference's own
    order -- 256-entry blocks from zero added one after the other, OtOr last, y as one chain with
    product and sum rounded separately (src/accel/als/implicit.rs:110-117) -- so the row stays
    within 1e-4 of the oracle although the ORACLE is further than that from float64."""
    from lkpy_amd import _device as D
    from lkpy_amd import _native

    rng = np.random.default_rng(19)
    k, n_cols, long_len = 320, 130_000, 120_000
    mat = _csr(rng, 24, n_cols, [long_len, 9000, 257, 256, 0, 1])
    other = (np.abs(rng.standard_normal((n_cols, k))) * 0.05).astype(np.float32)
    other[rng.random((n_cols, k)) < 0.3] *= -1.0
    pool = (np.abs(rng.standard_normal((2048, k))) * 0.05).astype(np.float32)
    pool[rng.random((2048, k)) < 0.3] *= -1.0
    rep = rng.random(n_cols) < 0.9
    other[rep] = pool[rng.integers(0, 2048, int(rep.sum()))]
    this = np.zeros((mat.shape[0], k), np.float32)
    otor = oracle.implicit_otor(other, 0.1)
    want = this.copy()
    oracle.als_half_epoch(mat, want, other, otor)
    exact, _ = oracle.als_referee_f64(mat, other, 0.1, with_cond=False)
    csr = D.DeviceCSR.from_arrays(mat.indptr.astype(np.int32), mat.indices, mat.data, mat.shape, gpu)
    plan = D.ALSPlan(csr, k, _native.SOLVER_AUTO)
    d_this = D.to_device_padded(this, gpu)
    import torch

    plan.half_epoch(d_this, D.to_device_padded(other, gpu),
                    torch.from_numpy(otor).to(gpu))  # the oracle's own OtOr: identical inputs
    plan.check_status()
    got = D.to_host_unpadded(d_this, k)
    nz = np.diff(mat.indptr) > 0
    rel = np.linalg.norm(got[nz].astype(np.float64) - want[nz], axis=1) / np.linalg.norm(want[nz], axis=1)
    o64 = np.linalg.norm(want[0].astype(np.float64) - exact[0]) / np.linalg.norm(exact[0])
    print(f"\nk=320, {long_len}-entry row: GPU vs oracle {rel[0]:.2e} (oracle vs float64 {o64:.2e}); "
          f"worst row {rel.max():.2e}")
    assert rel.max() < RTOL, rel
    assert rel[0] < 3e-5
