"""GPU parity: user-kNN -- scoring kernel vs the oracle's restatement of
src/accel/knn/user_score.rs, neighbour similarities vs SciPy, the component vs the reference's
golden predictions (tests/models/user-user-preds.csv)."""
from pathlib import Path

import numpy as np
import pandas as pd
import pyarrow as pa
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).parent / "golden"


@pytest.mark.parametrize("explicit", [True, False])
def test_uknn_score_kernel_vs_oracle(gpu, oracle, ml_small, rng, explicit):
    import torch

    from lkpy_amd import _device as D

    rmat = sps.csr_array(ml_small["rmat"])
    _uv, ur, _means = oracle.uknn_prepare(rmat, explicit)
    n_users, n_items = ur.shape
    drat = D.DeviceCSR(torch.from_numpy(ur.indptr.astype(np.int64)).to(gpu),
                       torch.from_numpy(ur.indices.astype(np.int32)).to(gpu),
                       torch.from_numpy(ur.data.astype(np.float32)).to(gpu) if explicit else None,
                       ur.shape, None)
    nq = 12
    nbr_ptr, nbr_rows, nbr_sims, tgt_ptr, tgt = [0], [], [], [0], []
    for q in range(nq):
        m = int(rng.integers(0, 200)) if q else 0  # query 0: no neighbours
        rows = np.sort(rng.choice(n_users, m, replace=False)).astype(np.int32)
        sims = rng.random(m).astype(np.float32) + 1e-3
        if m > 5:
            sims[3] = sims[4]  # a tie
        t = rng.choice(n_items, 150, replace=False).astype(np.int32)
        t[::17] = -1  # null targets
        nbr_rows.append(rows); nbr_sims.append(sims); tgt.append(t)
        nbr_ptr.append(nbr_ptr[-1] + m); tgt_ptr.append(tgt_ptr[-1] + len(t))
    to = lambda a, dt: torch.from_numpy(np.concatenate(a).astype(dt)).to(gpu)  # noqa: E731
    s, c = D.uknn_score_batch(drat, torch.tensor(nbr_ptr, device=gpu), to(nbr_rows, np.int32),
                              to(nbr_sims, np.float32), torch.tensor(tgt_ptr, device=gpu),
                              to(tgt, np.int32), 30, 2)
    s = s.cpu().numpy()
    for q in range(nq):
        want = oracle.uknn_score(ur, nbr_rows[q], nbr_sims[q], tgt[q], 30, 2, explicit)
        got = s[tgt_ptr[q]:tgt_ptr[q + 1]]
        assert np.array_equal(np.isnan(got), np.isnan(want)), q
        ok = ~np.isnan(want)
        # the same ScoreAccumulator steps, the same summation order: the oracle's bits
        assert np.array_equal(got[ok].view(np.uint32), want[ok].astype(np.float32).view(np.uint32)), q
    assert np.all(np.isnan(s[: tgt_ptr[1]]))  # no neighbours -> nothing scored


def test_csr_rows_dot_vs_scipy(gpu, oracle, ml_small, rng):
    import torch

    from lkpy_amd import _device as D

    uv, _ur, _m = oracle.uknn_prepare(sps.csr_array(ml_small["rmat"]), True)
    uv.sort_indices()
    X = np.zeros((uv.shape[1], 70), dtype=np.float32)  # two lane blocks of queries
    for b in range(70):
        r = int(rng.integers(0, uv.shape[0]))
        X[uv.indices[uv.indptr[r]:uv.indptr[r + 1]], b] = uv.data[uv.indptr[r]:uv.indptr[r + 1]]
    dcsr = D.DeviceCSR.from_arrays(uv.indptr, uv.indices, uv.data, uv.shape, gpu)
    got = D.csr_rows_dot(dcsr, torch.from_numpy(X).to(gpu)).cpu().numpy()
    want = (uv @ X).T
    assert got.shape == want.shape and np.allclose(got, want, rtol=1e-5, atol=1e-6)


def test_user_knn_component_golden(gpu, oracle, ml_small):
    """UserKNNScorer(k=30, min_sim=1e-6) through the component, batched: the reference's
    golden predictions (tolerance of its own test: 0.01) and the oracle tightly."""
    from lkpy_amd.data import ItemList, RecQuery, load_movielens_npz
    from lkpy_amd.knn import UserKNNScorer

    ds = load_movielens_npz(GOLDEN / "ml_small.npz")
    uknn = UserKNNScorer(k=30, min_sim=1.0e-6)
    uknn.train(ds)
    assert uknn.is_trained() and uknn.user_ratings.shape == (671, 9125)
    known = pd.read_csv(GOLDEN / "user-user-preds.csv")
    groups = list(known.groupby("user_id"))
    # like predict_pipeline: the history lookup supplies the user's training ratings
    queries = [RecQuery(user_id=int(u), user_items=ds.user_row(int(u))) for u, _ in groups]
    lists = [ItemList(g.item_id.values) for _, g in groups]
    res = uknn.score_batch(queries, lists)
    got = np.concatenate([r.scores() for r in res])
    exp = np.concatenate([g.prediction.values for _, g in groups])
    assert not np.any(np.isnan(got) & ~np.isnan(exp))
    err = np.abs(got - exp)
    err = err[~np.isnan(err)]
    assert err.max() < 0.01 and np.median(err) < 1e-5, (err.max(), np.median(err))
    # stored-vector path (no history), one query at a time == batched
    one = uknn(query=int(groups[0][0]), items=lists[0]).scores()
    assert np.allclose(one, res[0].scores(), rtol=1e-4, atol=1e-4, equal_nan=True)
    # unknown user without history / unknown items / empty item list
    assert np.all(np.isnan(uknn(query=-5, items=ItemList([1, 2])).scores()))
    r = uknn(query=int(groups[0][0]), items=ItemList([int(lists[0].ids()[0]), -77]))
    assert np.isnan(r.scores()[1])
    assert len(uknn(query=int(groups[0][0]), items=ItemList([])).scores()) == 0
    # implicit feedback: scores are sums of similarities, >= 0
    imp = UserKNNScorer(k=20, feedback="implicit")
    imp.train(ds)
    sc = imp(query=int(groups[1][0]), items=ItemList(ds.items.ids()[:300])).scores()
    assert np.all(sc[~np.isnan(sc)] > 0)


def test_user_score_seam_consumer_lines(gpu, oracle, ml_small):
    "src/lenskit/knn/user.py:196-254, line for line, against the `_accel.knn` stand-ins"
    from lkpy_amd import _accel
    from lkpy_amd.matrix import SparseRowArray

    knn = _accel.knn
    rmat = sps.csr_array(ml_small["rmat"])
    user_vectors, centred, means = oracle.uknn_prepare(rmat, True)
    user_ratings = SparseRowArray.from_scipy(centred, values=True)
    uidx, max_nbrs, min_nbrs, min_sim = 42, 30, 1, 1.0e-6
    ratings = user_vectors[[uidx], :].toarray()[0, :]
    umean = means.ravel()[uidx].item()
    # ---- user.py:196-251 ----
    nbr_sims = user_vectors @ ratings
    nbr_sims[uidx] = 0
    nbr_idxs = np.arange(user_vectors.shape[0], dtype=np.int32)
    nbr_mask = nbr_sims >= min_sim
    kn_sims = nbr_sims[nbr_mask]
    kn_idxs = nbr_idxs[nbr_mask]
    iidxs = np.concatenate([np.arange(0, 500), [-1]])
    ki_mask = iidxs >= 0
    usable_iidxs = pa.array(iidxs[ki_mask], pa.int32())
    kn_idxs = pa.array(kn_idxs, pa.int32())
    kn_sims = pa.array(kn_sims, pa.float32())
    scores = knn.user_score_items_explicit(usable_iidxs, kn_idxs, kn_sims, user_ratings,
                                           max_nbrs, min_nbrs)
    scores = scores.to_numpy(zero_copy_only=False, writable=True)
    scores += umean
    # ----
    want = oracle.uknn_score(centred, np.flatnonzero(nbr_mask), nbr_sims[nbr_mask],
                             iidxs[ki_mask], max_nbrs, min_nbrs, True) + np.float32(umean)
    assert np.array_equal(np.isnan(scores), np.isnan(want))
    ok = ~np.isnan(want)
    assert np.allclose(scores[ok], want[ok], rtol=2e-5, atol=1e-5)
    imp = knn.user_score_items_implicit(usable_iidxs, kn_idxs, kn_sims,
                                        SparseRowArray.from_scipy(centred, values=False),
                                        max_nbrs, min_nbrs)
    assert isinstance(imp, pa.FloatArray) and imp.null_count > 0


@pytest.mark.parametrize("diag", [True, False])
def test_fast_col_cooc_vs_scipy(gpu, rng, diag):
    "the reference's own check (tests/data/test_matrix.py:153-200): M^T M of a binary matrix"
    from lkpy_amd.matrix import fast_col_cooc

    mat = sps.random_array((300, 120), density=0.08, format="coo", rng=rng, dtype=np.float32)
    mat.data = np.ones(mat.nnz)
    cooc = (mat.T @ mat).toarray()
    if not diag:
        cooc[np.diag_indices(120)] = 0
    res = fast_col_cooc(mat.row.astype(np.int32), mat.col.astype(np.int32), mat.shape,
                        include_diagonal=diag)
    assert isinstance(res, sps.coo_array) and res.shape == (120, 120) and res.dtype == np.int32
    assert np.all(res.toarray() == cooc)
    dense = fast_col_cooc(pa.array(mat.row.astype(np.int32)), pa.array(mat.col.astype(np.int32)),
                          mat.shape, include_diagonal=diag, dense=True)
    assert isinstance(dense, np.ndarray) and np.all(dense == cooc)
    with pytest.raises(NotImplementedError):
        fast_col_cooc(mat.row, mat.col, mat.shape, ordered=True)
