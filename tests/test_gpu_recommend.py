"""
GPU: the ``als-implicit.toml`` recommend path for whole BATCHES of users:
training history -> fold-in -> scores of all items -> top-n without the history, which the
reference runs query by query (src/lenskit/batch/_runner.py:283-308 ->
src/lenskit/basic/history.py:77-95 -> src/lenskit/als/_implicit.py:77-130 ->
src/lenskit/als/_common.py:133-175 -> src/accel/data/sorting.rs:132-172).

* ``lk_csr_gather_rows`` against NumPy row slicing (both offset widths, unknown users, ratings
  scaled / constant / mean-centred);
* ml-latest-small: ``batch.recommend`` through the row-gather path == the per-query list path,
  bit for bit (same histories' CSR, same kernels), with unknown users, ``use_ratings`` and
  ``user_embeddings = "prefer"``;
* the ML-25M shape (cfg2): 10 000 sampled users from a model trained through the component --
  EVERY fold-in vector within the raw 1e-4 of the oracle's ``_train_new_row`` restatement, and
  the lists bit-identical to the oracle's score + exclusion + heap top-100 from the same vectors.
"""
from pathlib import Path

import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).parent / "golden"


@pytest.mark.parametrize("wide", [False, True])
def test_gather_rows_matches_numpy(gpu, wide):
    import torch

    from lkpy_amd import _device as D

    rng = np.random.default_rng(3)
    mat = sps.random(300, 500, density=0.05, format="csr", dtype=np.float32, random_state=7)
    mat.sort_indices()
    mat.data[:] = rng.integers(1, 11, mat.nnz).astype(np.float32) * 0.5
    ptr = mat.indptr.astype(np.int64 if wide else np.int32)
    csr = D.DeviceCSR.from_arrays(ptr, mat.indices, mat.data, mat.shape, gpu)
    rows = rng.integers(0, 300, 1000).astype(np.int32)
    rows[::17] = -1  # unknown users: empty rows
    bias = rng.standard_normal(500).astype(np.float32)
    d_bias = torch.from_numpy(bias).to(gpu)
    for scale, with_vals, cb in ((40.0, True, None), (1.0, True, d_bias), (3.0, False, None)):
        got = D.gather_rows(csr, rows, scale=scale, with_values=with_vals, col_bias=cb)
        lens = np.where(rows >= 0, np.diff(mat.indptr)[np.maximum(rows, 0)], 0)
        want_ptr = np.concatenate([[0], np.cumsum(lens)])
        assert got.indptr.dtype == torch.int64
        assert np.array_equal(got.indptr.cpu().numpy(), want_ptr)
        assert np.array_equal(got.h_indptr, want_ptr)
        sel = [np.arange(mat.indptr[r], mat.indptr[r + 1]) for r in rows if r >= 0]
        sel = np.concatenate(sel)
        assert np.array_equal(got.indices.cpu().numpy(), mat.indices[sel])
        if not with_vals:
            assert got.values is None
            continue
        v = mat.data[sel]
        if cb is not None:
            v = v - bias[mat.indices[sel]]
        want = v * np.float32(scale)
        assert np.array_equal(got.values.cpu().numpy().view(np.uint32), want.view(np.uint32))
    # no values in the matrix: the constant
    nov = D.DeviceCSR(csr.indptr, csr.indices, None, csr.shape, csr.h_indptr)
    got = D.gather_rows(nov, rows, scale=40.0)
    assert np.all(got.values.cpu().numpy() == np.float32(40.0))
    # an empty batch
    got = D.gather_rows(csr, np.zeros(0, np.int32))
    assert got.shape[0] == 0 and got.indices.numel() == 0


@pytest.fixture(scope="module")
def ml_ds():
    from lkpy_amd.data import load_movielens_npz

    return load_movielens_npz(GOLDEN / "ml_small.npz")


@pytest.mark.parametrize("mode", ["default", "use_ratings", "prefer"])
def test_batch_recommend_equals_per_query_path(gpu, ml_ds, mode):
    """``batch.recommend`` (HistoryBatch: rows gathered on the device) == ``recommend_batch``
    over the queries the lookup builds one by one: identical lists and score bits."""
    from lkpy_amd import batch
    from lkpy_amd.data import RecQuery
    from lkpy_amd.pipeline import Pipeline
    from lkpy_amd.training import TrainingOptions

    pipe = Pipeline.load_config(GOLDEN / "pipelines" / "als-implicit.toml")
    scorer = pipe.node("scorer").component
    scorer.config.epochs = 4
    if mode == "use_ratings":
        scorer.config.use_ratings = True
    if mode == "prefer":
        scorer.config.user_embeddings = "prefer"
    pipe.train(ml_ds, TrainingOptions(rng=42))
    lookup = pipe.node("history-lookup").component
    users = [int(u) for u in ml_ds.users.ids()[::7]] + [-5, 10**9]  # + two unknown users
    out = batch.recommend(pipe, users, 10)
    assert len(out) == len(users) and out.key_fields == ("user_id",)
    queries = [lookup(RecQuery.create(u)) for u in users]
    want_i, want_s = scorer.recommend_batch(queries, 10)
    hb = lookup.batch(users)
    assert len(hb) == len(users) and hb.user_nums[-1] == -1 and hb.lengths[-1] == 0
    got_i, got_s = scorer.recommend_batch(hb, 10)
    assert np.array_equal(got_i, want_i)
    assert np.array_equal(np.ascontiguousarray(got_s).view(np.uint32),
                          np.ascontiguousarray(want_s).view(np.uint32))
    # unknown users: nothing to recommend from (no history, no stored row)
    assert (got_i[-2:] == -1).all() and np.isnan(got_s[-2:]).all()
    assert len(out.lookup(-5)) == 0 and len(out.lookup(user_id=10**9)) == 0
    # the collection's lists are those rows; a user's own items never appear
    for pos in (0, 5, len(users) - 3):
        key, il = out[pos]
        assert key.user_id == users[pos] and il.ordered and len(il) == 10
        assert np.array_equal(il.numbers(vocabulary=scorer.items), got_i[pos])
        hist = ml_ds.user_row(users[pos]).numbers(vocabulary=scorer.items)
        assert not np.isin(got_i[pos], hist).any()
        # and the pipeline run for that one user gives the same list
        # (score bits: equal scores of different items may be listed in another order)
        one = pipe.run("recommender", query=users[pos], n=10)
        assert np.array_equal(np.asarray(one.scores(), np.float32).view(np.uint32),
                              np.ascontiguousarray(got_s[pos]).view(np.uint32))
    df = out.to_df()
    assert len(df) == out.total_items() == 10 * (len(users) - 2)
    assert list(df.columns[:4]) == ["user_id", "item_id", "score", "rank"]
    # device_output: the same arrays, nothing downloaded
    d_i, d_s = scorer.recommend_batch(hb, 10, device_output=True)
    assert d_i.is_cuda and np.array_equal(d_i.cpu().numpy(), got_i)


def test_recommend_batch_at_ml25m_shape(gpu, oracle):
    """cfg2 scale: 10 000 sampled users of a model trained through the component on the
    ML-25M-shaped synthetic (k = 64, 20 epochs, weight 40): every fold-in vector within the RAW
    1e-4 of the oracle's (``_train_new_row``: NumPy + SciPy cho_factor, _implicit.py:101-130), the
    top-100 lists bit-identical to the oracle's from the same vectors (ties by item number
    counted apart, as in bench.py), the histories excluded, and the host's share of the call
    printed."""
    import time

    import torch

    from lkpy_amd import _device as D
    from lkpy_amd import synth
    from lkpy_amd.als import ImplicitMFScorer
    from lkpy_amd.basic import UserTrainingHistoryLookup
    from lkpy_amd.data import Dataset, Vocabulary
    from lkpy_amd.training import TrainingOptions
    from oracle import parity

    ratings = synth.ml25m_like()
    n_users, n_items = ratings.shape
    rows = np.repeat(np.arange(n_users, dtype=np.int32), np.diff(ratings.indptr))
    ds = Dataset(Vocabulary(np.arange(n_users), "user", reorder=False),
                 Vocabulary(np.arange(n_items), "item", reorder=False),
                 rows, ratings.indices, {"rating": ratings.data})
    k, n = 64, 100
    scorer = ImplicitMFScorer(embedding_size=k, epochs=20, weight=40.0)
    scorer.train(ds, TrainingOptions(rng=42))
    lookup = UserTrainingHistoryLookup()
    lookup.train(ds)
    users = np.random.default_rng(11).choice(ds.users.ids(), 10000, replace=False)
    scorer.recommend_batch(lookup.batch(users[:256]), n)  # uploads
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hb = lookup.batch(users)
    got_i, got_s = scorer.recommend_batch(hb, n)
    wall = time.perf_counter() - t0
    assert got_i.shape == (10000, n) and (got_i >= 0).all()

    u_dev, valid, hist = scorer._history_batch_embeddings(hb)
    assert valid.all()
    u_gpu = D.to_host_unpadded(u_dev, k)
    Q, OtOr = scorer.item_embeddings, scorer._OtOr
    hp, cols = ds._indptr, ds._cols
    want_u = np.zeros_like(u_gpu)
    for r, un in enumerate(hb.user_nums):
        items = cols[hp[un]:hp[un + 1]]
        want_u[r] = oracle.als_fold_in(items, np.full(len(items), 40.0, np.float32), Q, OtOr)
    rel = np.linalg.norm(u_gpu.astype(np.float64) - want_u, axis=1) / \
        np.linalg.norm(want_u.astype(np.float64), axis=1)
    print(f"\ncfg2 batch recommend: 10 000 users in {wall * 1e3:.1f} ms; fold-in vs oracle: max "
          f"{rel.max():.2e}, median {np.median(rel):.2e}, rows over 1e-4: {(rel > 1e-4).sum()}")
    assert (rel <= 1e-4).all(), (int((rel > 1e-4).sum()), float(rel.max()))

    # the histories as the device gathered them == the training rows
    m = 2048
    ptr = np.zeros(m + 1, np.int64)
    np.cumsum(hb.lengths[:m], out=ptr[1:])
    ex = np.concatenate([cols[hp[un]:hp[un + 1]] for un in hb.user_nums[:m]]).astype(np.int32)
    assert np.array_equal(hist.indices[:ptr[m]].cpu().numpy(), ex)
    want_i, want_s = oracle.score_topn_batch(Q, u_gpu[:m], n, ptr, ex)
    acc = parity.topn_accounting(got_i[:m], got_s[:m], want_i, want_s,
                                 lambda r: oracle.score_dense(Q, u_gpu[r]))
    print("  lists vs the oracle from the same vectors:", acc)
    assert acc["ok"] and acc["score_rows_bit_identical"] and acc["mismatched_users"] == 0, acc
    assert acc["lists_identical"] >= m - 64, acc  # (ties among bit-equal scores only)
    for r in range(0, m, 97):
        assert not np.isin(got_i[r], ex[ptr[r]:ptr[r + 1]]).any()
