"""
``lk_iknn_recommend`` (csrc/iknn_recommend.hip): item-kNN "score every item + top-N" for a batch
of queries against the oracle's restatement of the reference pipeline
(src/lenskit/knn/item.py:231-295 -> src/accel/knn/item_score.rs:23-111 + accum.rs;
src/lenskit/basic/candidates.py:77-94; src/lenskit/basic/topn.py:45-69).

Bar: the SCORE of every listed item carries the reference accumulator's bits; the sorted score
rows of the lists are bit-identical to the oracle's; every listed item is a genuine candidate
(scored, not one of the query's own); which of several items with IDENTICAL score bits is listed
first / kept at the cut is the reference heap's sift order (unspecified: SURVEY 8g-8) and is
counted, not hidden.
"""
import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu


def _to(a, dev):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _model(oracle, rng, n_users, n_items, mean_len, save_nbrs):
    lens = np.clip(rng.geometric(1.0 / mean_len, n_users), 1, n_items // 2)
    lens[:3] = [n_items // 2, 700, 300]
    rows = np.repeat(np.arange(n_users), lens)
    cols = np.concatenate([rng.choice(n_items, ln, replace=False) for ln in lens])
    vals = rng.integers(1, 11, len(rows)).astype(np.float32) * 0.5
    rmat = sps.coo_array((vals, (rows, cols)), shape=(n_users, n_items))
    ui, iu, means, _ = oracle.iknn_prepare(rmat, True)
    sims = oracle.iknn_build(ui, iu, 1.0e-6, save_nbrs)
    return sps.csr_array(rmat), sims, np.asarray(means, dtype=np.float32).ravel()


def _queries(csr, means, users, rng, with_null=True):
    lens = np.diff(csr.indptr)[users]
    ptr = np.zeros(len(users) + 1, np.int64)
    np.cumsum(lens, out=ptr[1:])
    take = np.concatenate([np.arange(csr.indptr[u], csr.indptr[u + 1]) for u in users])
    idx = csr.indices[take].astype(np.int32)
    # history order as a query hands it over: NOT sorted by item (the accumulator's sums depend on it)
    for q in range(len(users)):
        seg = slice(ptr[q], ptr[q + 1])
        p = rng.permutation(ptr[q + 1] - ptr[q])
        idx[seg] = idx[seg][p]
        take[seg] = take[seg][p]
    val = (csr.data[take] - means[idx]).astype(np.float32)
    if with_null and len(idx) > 10:
        idx[5] = -1  # an unknown history item: skipped
    return ptr, idx, val


def _check(gi, gs, wi, ws, rows, ptr, idx):
    assert np.array_equal(gs.view(np.uint32), ws.view(np.uint32))  # sorted score rows, bit for bit
    ties = 0
    for q in range(len(gi)):
        g = gi[q][gi[q] >= 0]
        assert len(g) == int((wi[q] >= 0).sum())
        assert np.array_equal(rows[q][g].view(np.uint32), gs[q][: len(g)].view(np.uint32))
        own = idx[ptr[q]:ptr[q + 1]]
        assert not np.isin(g, own[own >= 0]).any()
        assert len(np.unique(g)) == len(g)
        if not np.array_equal(gi[q], wi[q]):
            ties += 1
    return ties


@pytest.mark.parametrize("walk", ["acc", "packed", "pieces", "acc-small-panel"])
@pytest.mark.parametrize("explicit", [True, False])
@pytest.mark.parametrize("save_nbrs,max_nbrs,n", [(50, 20, 10), (None, 5, 100), (None, 64, 30)])
def test_recommend_matches_the_reference_pipeline(gpu, oracle, monkeypatch, walk, explicit,
                                                  save_nbrs, max_nbrs, n):
    """The three kernels behind ``lk_iknn_recommend`` give the same bits: the ACCUMULATING kernel
    (round 6, the default: every weight is added to its target's LDS cell by ``ds_add_f32`` --
    same-address adds of an instruction applied in lane order, probed on the device --, targets
    beyond ``max_nbrs`` hits gathered and replayed by their own kernels), the list kernel with the
    (row x window) pieces PACKED 64 entries to the instruction (round 5; ``LK_REC_ACC=0``), and the
    piece-by-piece walk that one falls back to (``LK_REC_PACKED=0``)."""
    from lkpy_amd import _device as D
    from lkpy_amd import _native

    if walk == "pieces":
        monkeypatch.setenv("LK_REC_PACKED", "0")
    if walk == "packed":
        monkeypatch.setenv("LK_REC_ACC", "0")
    if walk.endswith("small-panel"):
        # the score panel's byte budget (LK_REC_PANEL_GB; round 6): so small here that the 124
        # queries go through in two batches of 64 rows -- the same lists
        monkeypatch.setenv("LK_REC_PANEL_GB", "0.000001")
        walk = "acc"

    rng = np.random.default_rng(7)
    n_users, n_items = 600, 9001  # three windows of 4096, the last one partial
    csr, sims, means = _model(oracle, rng, n_users, n_items, 25, save_nbrs)
    users = np.concatenate([[0, 1, 2], rng.choice(n_users, 120, replace=False)])
    ptr, idx, val = _queries(csr, means, users, rng)
    # one query without history
    ptr = np.concatenate([ptr, [ptr[-1]]])
    counts = np.diff(sims.indptr).astype(np.int64)
    per = np.where(idx >= 0, counts[np.maximum(idx, 0)], 0)
    cs = np.concatenate([[0], np.cumsum(per)])
    hits = cs[ptr[1:]] - cs[ptr[:-1]]
    dsims = D.DeviceCSR.from_arrays(sims.indptr.astype(np.int64), sims.indices, sims.data,
                                    sims.shape, gpu)
    gi, gs = D.iknn_recommend(dsims, _to(ptr, gpu), _to(idx, gpu),
                              _to(val, gpu) if explicit else None,
                              _to(means, gpu) if explicit else None, max_nbrs, 2, n, hits)
    gi, gs = gi.cpu().numpy(), gs.cpu().numpy()
    # the probes of the LDS atomic order must pass on an MI355X: the accumulating kernel is the
    # product path
    assert _native.load().lk_iknn_recommend_last_packed() == {"acc": 2, "packed": 1, "pieces": 0}[walk]
    wi, ws, rows = oracle.iknn_recommend_batch(sims, ptr, idx, val if explicit else None,
                                               means if explicit else None, max_nbrs, 2, n)
    ties = _check(gi, gs, wi, ws, rows, ptr, idx)
    assert (gi[-1] == -1).all() and np.isnan(gs[-1]).all()  # no history: an empty list
    # the heap path (more than max_nbrs hits on a target) was exercised by the long histories
    print(f"\nexplicit={explicit} save_nbrs={save_nbrs} max_nbrs={max_nbrs}: {len(gi)} queries, "
          f"{int((gi >= 0).sum())} listed items, lists differing among equal scores: {ties}")
    # (ties are the rule on this small integer-rated matrix: many items share a score exactly;
    # _check has established that every list is a genuine top-n of the oracle's score row)


@pytest.mark.parametrize("kernel", ["acc", "lists"])
@pytest.mark.parametrize("case", ["keep-own", "full-ranking", "min-nbrs-3", "wide-heaps", "one-batch"])
def test_recommend_corners_of_the_accumulating_kernel(gpu, oracle, monkeypatch, kernel, case):
    """Corners of ``iknn_score_acc_kernel`` and its helpers, each against the oracle and under both
    kernels: the own items kept (``exclude_refs = False``: nothing is marked in the count cells),
    the full ranking (``n = -1``: no class maxima, the sort path), ``min_nbrs = 3`` (cells of
    targets with one or two hits start as NaN), ``max_nbrs`` in the thousands (the replay kernel's
    heaps beyond 64 KiB of LDS; hardly any target is queued), and a call of ONE batch (no side
    stream, no second panel)."""
    from lkpy_amd import _device as D
    from lkpy_amd import _native

    if kernel == "lists":
        monkeypatch.setenv("LK_REC_ACC", "0")
    rng = np.random.default_rng(11)
    n_users, n_items = 500, 8300
    csr, sims, means = _model(oracle, rng, n_users, n_items, 30, None)
    users = np.concatenate([[0, 1, 2], rng.choice(n_users, 40 if case == "one-batch" else 90,
                                                   replace=False)])
    ptr, idx, val = _queries(csr, means, users, rng)
    max_nbrs, min_nbrs, n, exclude = 20, 1, 25, True
    if case == "keep-own":
        exclude = False
    elif case == "full-ranking":
        n = -1
    elif case == "min-nbrs-3":
        max_nbrs, min_nbrs = 6, 3
    elif case == "wide-heaps":
        max_nbrs = 2500
    if case != "one-batch":
        monkeypatch.setenv("LK_REC_PANEL_ROWS", "64")  # two batches: the side stream, both panels
    counts = np.diff(sims.indptr).astype(np.int64)
    per = np.where(idx >= 0, counts[np.maximum(idx, 0)], 0)
    cs = np.concatenate([[0], np.cumsum(per)])
    hits = cs[ptr[1:]] - cs[ptr[:-1]]
    dsims = D.DeviceCSR.from_arrays(sims.indptr.astype(np.int64), sims.indices, sims.data,
                                    sims.shape, gpu)
    gi, gs = D.iknn_recommend(dsims, _to(ptr, gpu), _to(idx, gpu), _to(val, gpu), _to(means, gpu),
                              max_nbrs, min_nbrs, n, hits, exclude)
    gi, gs = gi.cpu().numpy(), gs.cpu().numpy()
    assert _native.load().lk_iknn_recommend_last_packed() == (2 if kernel == "acc" else 1)
    cols = n_items if n < 0 else n
    wi, ws, rows = oracle.iknn_recommend_batch(sims, ptr, idx, val, means, max_nbrs, min_nbrs, cols,
                                               exclude_refs=exclude)
    assert np.array_equal(gs.view(np.uint32), ws.view(np.uint32))  # sorted score rows, bit for bit
    for q in range(len(gi)):
        g = gi[q][gi[q] >= 0]
        assert len(g) == int((wi[q] >= 0).sum()) and len(np.unique(g)) == len(g)
        assert np.array_equal(rows[q][g].view(np.uint32), gs[q][: len(g)].view(np.uint32))
        own = idx[ptr[q]:ptr[q + 1]]
        if exclude:
            assert not np.isin(g, own[own >= 0]).any()
    if case == "keep-own":  # the own items do appear (a user's own items score high)
        assert any(np.isin(gi[q][gi[q] >= 0], idx[ptr[q]:ptr[q + 1]]).any() for q in range(len(gi)))


def test_recommend_through_the_scorer_and_batch_runner(gpu, oracle, ml_small):
    """``ItemKNNScorer.recommend_batch`` / ``batch.recommend`` on ml-latest-small: the same lists
    as one ``pipe.run('recommender')`` per user (the scorer's own per-query path), scores bit for
    bit, and never a per-user pipeline run inside the batch call."""
    from lkpy_amd import batch
    from lkpy_amd.data import load_movielens_npz
    from lkpy_amd.knn import ItemKNNScorer
    from lkpy_amd.pipeline import topn_pipeline
    from pathlib import Path

    ds = load_movielens_npz(Path(__file__).parent / "golden" / "ml_small.npz")
    scorer = ItemKNNScorer(max_nbrs=20, min_nbrs=2, save_nbrs=200)
    pipe = topn_pipeline(scorer, predicts_ratings=True, n=10)
    pipe.train(ds)
    users = list(ds.users._ids[:40]) + [int(ds.users._ids[-1])]
    calls = []
    orig = pipe.run
    pipe.run = lambda *a, **k: (calls.append(a), orig(*a, **k))[1]
    got = batch.recommend(pipe, users, 10)
    assert not calls, "batch.recommend must not fall back to one pipeline run per user"
    assert got.key_fields == ("user_id",) and len(got) == len(users)
    assert got.lookup(user_id=users[0]) is got.lookup((users[0],))
    df = got.to_df()
    assert list(df.columns[:2]) == ["user_id", "item_id"] and "rank" in df.columns
    pipe.run = orig
    diff = 0
    for u in users:
        want = pipe.run("recommender", query=u, n=10)
        g = got.lookup(u)  # an ItemListCollection keyed by user_id, like the reference's
        ws_, gs_ = np.asarray(want.scores(), np.float32), np.asarray(g.scores(), np.float32)
        assert np.array_equal(ws_.view(np.uint32), gs_.view(np.uint32)), u
        if not np.array_equal(want.numbers(vocabulary=scorer.items),
                              g.numbers(vocabulary=scorer.items)):
            diff += 1
    assert diff <= 4  # equal-score items only (checked bit for bit above)
    # round 6: the batch went by USER NUMBER (HistoryBatch: histories gathered from the
    # HBM-resident training matrix, mean-centred in the gather kernel, hit counts per user from
    # the device) -- the same arrays as the per-query list path, bit for bit, unknown user included
    from lkpy_amd.data import RecQuery

    lookup = pipe.node("history-lookup").component
    ids = users + [-7]
    gi, gs = scorer.recommend_batch(lookup.batch(ids), 10)
    li, ls = scorer.recommend_batch([lookup(RecQuery.create(u)) for u in ids], 10)
    assert np.array_equal(gi, li)
    assert np.array_equal(np.ascontiguousarray(gs).view(np.uint32),
                          np.ascontiguousarray(ls).view(np.uint32))
    assert (gi[-1] == -1).all() and np.isnan(gs[-1]).all()
    imp = ItemKNNScorer(max_nbrs=20, min_nbrs=2, save_nbrs=200, feedback="implicit")
    pipe2 = topn_pipeline(imp, n=10)
    pipe2.train(ds)
    lk2 = pipe2.node("history-lookup").component
    gi, gs = imp.recommend_batch(lk2.batch(ids), 10)
    li, ls = imp.recommend_batch([lk2(RecQuery.create(u)) for u in ids], 10)
    assert np.array_equal(gi, li)
    assert np.array_equal(np.ascontiguousarray(gs).view(np.uint32),
                          np.ascontiguousarray(ls).view(np.uint32))


def test_nan_similarity_is_the_reference_error(gpu):
    from lkpy_amd import _device as D
    from lkpy_amd import _native

    sims = sps.csr_array(np.array([[0, np.nan, 0.5], [0.2, 0, 0.1], [0.5, 0.1, 0]], np.float32))
    dsims = D.DeviceCSR.from_arrays(sims.indptr.astype(np.int64), sims.indices, sims.data,
                                    sims.shape, gpu)
    ptr = np.array([0, 2], np.int64)
    idx = np.array([0, 1], np.int32)
    with pytest.raises(ValueError, match="similarity is null"):
        D.iknn_recommend(dsims, _to(ptr, gpu), _to(idx, gpu), None, None, 5, 1, 2,
                         np.array([4], np.int64))
    assert _native is not None


def test_history_batch_equals_list_path_at_ml25m_shape(gpu):
    """``iknn-explicit.toml``'s scorer at the ML-25M shape (``save_nbrs = max_nbrs = 100``), 5 000
    sampled users: ``recommend_batch`` by USER NUMBER (``HistoryBatch``: histories gathered and
    mean-centred on the device, hit counts per user from one device pass, two batches of the
    recommend kernel) against the per-query list path (one ``RecQuery`` per user, host arrays) --
    the same index and score arrays, bit for bit.  What the array-level bench leg and the
    ml-latest-small test leave open between them: histories of up to 32 202 entries, more
    queries than one panel batch holds, the heaviest-first reordering undone."""
    from lkpy_amd import synth
    from lkpy_amd.basic import UserTrainingHistoryLookup
    from lkpy_amd.data import Dataset, RecQuery, Vocabulary
    from lkpy_amd.knn import ItemKNNScorer

    ratings = synth.ml25m_like()
    n_u, n_i = ratings.shape
    ds = Dataset(Vocabulary(np.arange(n_u), "user", reorder=False),
                 Vocabulary(np.arange(n_i), "item", reorder=False),
                 np.repeat(np.arange(n_u, dtype=np.int32), np.diff(ratings.indptr)),
                 ratings.indices, {"rating": ratings.data})
    scorer = ItemKNNScorer(max_nbrs=100, min_nbrs=1, save_nbrs=100)
    scorer.train(ds)
    lookup = UserTrainingHistoryLookup()
    lookup.train(ds)
    rng = np.random.default_rng(43)
    users = rng.choice(n_u, 5000, replace=False)
    users[:3] = np.argsort(-np.diff(ratings.indptr))[:3]  # the three longest histories
    gi, gs = scorer.recommend_batch(lookup.batch(users), 100)
    li, ls = scorer.recommend_batch([lookup(RecQuery.create(int(u))) for u in users], 100)
    assert gi.shape == (5000, 100) and (gi >= 0).all()
    assert np.array_equal(gi, li)
    assert np.array_equal(np.ascontiguousarray(gs).view(np.uint32),
                          np.ascontiguousarray(ls).view(np.uint32))
    for r in (0, 1, 2, 77):  # never one of the user's own items
        own = ratings.indices[ratings.indptr[users[r]]:ratings.indptr[users[r] + 1]]
        assert not np.isin(gi[r], own).any()
