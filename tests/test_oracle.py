"""
CPU tests of the ORACLE itself: it is pinned against the reference's own golden
vectors, closed forms and unit tests before any GPU result is compared with it.
All citations relative to the reference checkout (lenskit/lkpy).
"""
import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sps
from pathlib import Path

GOLDEN = Path(__file__).parent / "golden"


# ---- item-kNN: pinned by tests/models/item-item-preds.csv ---------------------------


@pytest.fixture(scope="module")
def iknn_model(oracle, ml_small):
    ui, iu, means, _ = oracle.iknn_prepare(ml_small["rmat"], True)
    sims = oracle.iknn_build(ui, iu, 1.0e-6, None)
    return ui, iu, means, sims


def test_iknn_known_preds(oracle, ml_small, iknn_model):
    """
    ``test_ii_known_preds`` (tests/models/test_knn_item_item.py:413-453):
    ItemKNNScorer(k=20, min_sim=1e-6) on ml-latest-small must reproduce the 1288 golden
    predictions.  The reference's only hard assertion is "no erroneously missing
    prediction" (line 435); its error histogram allows a few tie-induced outliers.
    """
    _ui, _iu, means, sims = iknn_model
    known = pd.read_csv(GOLDEN / "item-item-preds.csv")
    assert len(known) == 1288
    csr = sps.csr_array(ml_small["rmat"])
    errs, missing = [], 0
    for uid, grp in known.groupby("user_id"):
        u = int(np.searchsorted(ml_small["user_ids"], uid))
        hist = csr.indices[csr.indptr[u] : csr.indptr[u + 1]]
        rates = csr.data[csr.indptr[u] : csr.indptr[u + 1]].astype(np.float32) - means[hist]
        tg = np.searchsorted(ml_small["item_ids"], grp.item_id.values).astype(np.int32)
        sc, cnt = oracle.iknn_score(sims, hist, rates, tg, 20, 1)
        sc = sc + means[tg]
        missing += int(np.sum(np.isnan(sc) & ~np.isnan(grp.prediction.values)))
        e = grp.prediction.values - sc
        errs.extend(e[~np.isnan(e)])
    errs = np.abs(np.array(errs))
    assert missing == 0
    assert len(errs) == 1288
    assert np.sum(errs > 1e-5) <= 5  # top-20 boundary ties (the golden file is from Java LensKit)
    assert np.median(errs) < 1e-6


def test_iknn_sims_match_dense_cosine(oracle, ml_small, iknn_model):
    "sims == centred cosine computed densely in float64 (cf. test_knn_item_item.py:331-338)."
    ui, _iu, _means, sims = iknn_model
    rng = np.random.default_rng(1)
    dense = ui.astype(np.float64)
    for i in rng.choice(np.flatnonzero(np.diff(sims.indptr) > 0), 40):
        cols = sims.indices[sims.indptr[i] : sims.indptr[i + 1]]
        vals = sims.data[sims.indptr[i] : sims.indptr[i + 1]]
        ref = np.asarray((dense[:, [i]].T @ dense[:, cols]).todense()).ravel()
        assert np.allclose(vals, ref, atol=1e-6)
    assert sims.data.min() > 0 and sims.data.max() <= 1 + 1e-6
    assert not np.any(sims.diagonal() != 0)


def test_iknn_toy_closed_form(oracle):
    "tests/models/test_knn_item_item.py:106-162: sim(6,7) on the 14-rating toy set."
    recs = [(1, 6, 4.0), (2, 6, 2.0), (1, 7, 3.0), (2, 7, 2.0), (3, 7, 5.0), (4, 7, 2.0),
            (1, 8, 3.0), (2, 8, 4.0), (3, 8, 3.0), (4, 8, 2.0), (5, 8, 3.0), (6, 8, 2.0),
            (1, 9, 3.0), (3, 9, 4.0)]  # fmt: skip
    u = np.array([r[0] - 1 for r in recs])
    i = np.array([r[1] - 6 for r in recs])
    v = np.array([r[2] for r in recs], np.float32)
    ui, iu, means, all_zero = oracle.iknn_prepare(sps.coo_array((v, (u, i)), shape=(6, 4)), True)
    assert not all_zero
    assert np.allclose(means, [3.0, 3.0, 17 / 6, 3.5])
    for save in (None, 500):
        S = oracle.iknn_build(ui, iu, 1e-6, save).toarray()
        six = np.array([4.0, 2.0]) - 3.0
        seven = np.array([3.0, 2.0, 5.0, 2.0]) - 3.0
        num = six[0] * seven[0] + six[1] * seven[1]
        assert S[0, 1] == pytest.approx(num / (np.linalg.norm(six) * np.linalg.norm(seven)), rel=1e-5)


def test_iknn_constant_ratings_flagged(oracle):
    "test_ii_warns_center (test_knn_item_item.py:211-216): all-equal ratings centre to zero."
    rmat = sps.coo_array((np.ones(6, np.float32), ([0, 0, 1, 1, 2, 2], [0, 1, 0, 1, 0, 1])), (3, 2))
    _ui, _iu, _m, all_zero = oracle.iknn_prepare(rmat, True)
    assert all_zero


def test_iknn_save_nbrs_truncation(oracle, ml_small, iknn_model):
    "test_ii_large_models (test_knn_item_item.py:256-370): bounded rows are the top of unbounded."
    ui, iu, _means, full = iknn_model
    lim = oracle.iknn_build(ui, iu, 1.0e-6, 100)
    assert np.all(np.diff(lim.indptr) <= 100)
    rng = np.random.default_rng(3)
    for i in rng.choice(full.shape[0], 60):
        fc = full.indices[full.indptr[i] : full.indptr[i + 1]]
        fv = full.data[full.indptr[i] : full.indptr[i + 1]]
        lc = lim.indices[lim.indptr[i] : lim.indptr[i + 1]]
        lv = lim.data[lim.indptr[i] : lim.indptr[i + 1]]
        assert np.all(np.diff(lc) > 0)
        assert np.all(np.isin(lc, fc))
        if len(fc) <= 100:
            assert np.array_equal(lc, fc) and np.array_equal(lv, fv)
        else:
            assert len(lc) == 100
            kth = np.sort(fv)[-100]
            assert lv.min() == kth
            assert set(fc[fv > kth]) <= set(lc)


def test_iknn_score_implicit_sum_of_topk(oracle, ml_small):
    "test_ii_implicit_large (test_knn_item_item.py:373-410): score == sum of the k largest sims."
    rmat = ml_small["rmat"]
    ind = sps.coo_array((np.ones(rmat.nnz, np.float32), (rmat.row, rmat.col)), rmat.shape)
    ui, iu, _m, _ = oracle.iknn_prepare(ind, False)
    sims = oracle.iknn_build(ui, iu, 1e-6, None)
    dense = sims.toarray()
    csr = sps.csr_array(ind)
    rng = np.random.default_rng(5)
    for u in rng.choice(rmat.shape[0], 10):
        hist = csr.indices[csr.indptr[u] : csr.indptr[u + 1]]
        tg = rng.choice(rmat.shape[1], 50, replace=False).astype(np.int32)
        sc, cnt = oracle.iknn_score(sims, hist, None, tg, 5, 1)
        for t, s, c in zip(tg, sc, cnt):
            col = dense[hist, t]
            col = np.sort(col[col > 0])[::-1][:5]
            assert c == len(col)
            if len(col) == 0:
                assert np.isnan(s)
            else:
                assert s == pytest.approx(col.sum(), rel=1e-5)


def test_iknn_score_nulls_and_min_nbrs(oracle, iknn_model):
    _ui, _iu, means, sims = iknn_model
    hist = np.array([0, 5, -1, 30], np.int32)  # -1: unknown history item, skipped
    rates = np.array([1.0, -0.5, 9.0, 0.25], np.float32)
    tg = np.array([1, -1, 2, 9000], np.int32)
    sc, cnt = oracle.iknn_score(sims, hist, rates, tg, 20, 1)
    assert np.isnan(sc[1]) and cnt[1] == -1  # null target
    sc2, cnt2 = oracle.iknn_score(sims, hist, rates, tg, 20, 4)
    assert np.all(np.isnan(sc2[cnt2 < 4]))
    assert np.array_equal(cnt, cnt2)


# ---- top-N: pinned by src/accel/indirect/heap.rs:105-162, tests/accel/test_argsort.py ----


def test_argtopn_rust_unit_vectors(oracle):
    assert len(oracle.argtopn(np.array([], np.float32), 5)) == 0  # test_heap_empty
    assert oracle.argtopn(np.array([10.0], np.float32), 5).tolist() == [0]  # test_heap_one
    assert oracle.argtopn(np.array([10.0, 20.0], np.float32), 5).tolist() == [1, 0]  # test_heap_two
    s = np.arange(1, 11, dtype=np.float32)  # test_heap_sort
    assert oracle.argtopn(s, 5).tolist() == [9, 8, 7, 6, 5]


@pytest.mark.parametrize("seed", range(20))
def test_argtopn_properties(oracle, seed):
    "tests/accel/test_argsort.py:60-211: length, descending, nothing excluded beats the min, NaN skipped."
    rng = np.random.default_rng(seed)
    n_s = int(rng.integers(1, 3000))
    s = rng.standard_normal(n_s).astype(np.float32)
    s[rng.random(n_s) < 0.1] = np.nan
    if seed % 3 == 0:
        s = np.round(s, 1)  # ties
    n = int(rng.integers(1, 200))
    top = oracle.argtopn(s, n)
    valid = ~np.isnan(s)
    assert len(top) == min(n, valid.sum())
    assert not np.any(np.isnan(s[top]))
    assert np.all(np.diff(s[top]) <= 0)
    if len(top):
        rest = np.setdiff1d(np.flatnonzero(valid), top)
        assert not np.any(s[rest] > s[top].min())
    assert len(set(top.tolist())) == len(top)
    full = oracle.argsort_descending(s)
    assert len(full) == valid.sum() and np.all(np.diff(s[full]) <= 0)


# ---- implicit ALS: behavioural pins (tests/models/test_als_implicit.py) ------------------


def test_als_row_matches_float64(oracle, rng):
    n_rows, n_cols, k = 200, 300, 16
    mat = sps.random(n_rows, n_cols, 0.05, random_state=1, format="csr", dtype=np.float32)
    mat.data[:] = 40.0
    mat = sps.csr_array(mat)
    other = (rng.standard_normal((n_cols, k)) * 0.3).astype(np.float32)
    this = np.zeros((n_rows, k), np.float32)
    otor = oracle.implicit_otor(other, 0.1)
    frob = oracle.als_half_epoch(mat, this, other, otor)
    exact = oracle.als_half_epoch_f64(mat, other, 0.1)
    assert np.linalg.norm(this - exact) / np.linalg.norm(exact) < 1e-5
    assert frob == pytest.approx(np.linalg.norm(exact), rel=1e-4)
    empty = np.diff(mat.indptr) == 0
    assert np.all(this[empty] == 0)  # implicit.rs:98-101


def test_als_train_ml_small_behaviour(oracle, ml_small):
    """
    tests/models/test_als_implicit.py:327-360 (shapes after training on ml-latest-small),
    109-120 (finite predictions), 139-218 (fold-in ~= trained embedding).
    """
    rmat = ml_small["rmat"]
    ind = sps.coo_array((np.ones(rmat.nnz, np.float32), (rmat.row, rmat.col)), rmat.shape)
    st = oracle.als_train(ind, 25, 5, np.random.SeedSequence(42).spawn(3)[2])
    P, Q = st.user_embeddings, st.item_embeddings
    assert P.shape == (671, 25) and Q.shape == (9125, 25)
    assert np.all(np.isfinite(P)) and np.all(np.isfinite(Q))
    dP = [d[0] for d in st.deltas]
    assert dP[-1] < dP[0]  # converging
    empty = np.bincount(ind.col, minlength=9125) == 0
    assert empty.sum() == 9125 - 9066 and np.all(Q[empty] == 0)
    # fold-in (the Python/SciPy path, _implicit.py:101-130) == one more user half-epoch row
    # (the Rust/sposv path) against the same Q and OtOr; and it stays near the trained row
    # (tests/models/test_als_implicit.py:139-218 allow 0.1 abs after full training)
    csr = sps.csr_array(oracle.als_prepare_matrix(ind, 40.0))
    csr.sort_indices()
    P2 = P.copy()
    oracle.als_half_epoch(csr, P2, Q, st.OtOr)
    for u in (0, 10, 100):
        items = csr.indices[csr.indptr[u] : csr.indptr[u + 1]]
        x = oracle.als_fold_in(items, np.full(len(items), 40.0, np.float32), Q, st.OtOr)
        assert np.abs(x - P2[u]).max() < 2e-4 * max(1.0, np.abs(P2[u]).max())
        assert np.abs(x - P[u]).max() < 1.0  # only 5 epochs here: near, not converged
    s = oracle.score_dense(Q, P[0])
    assert np.allclose(s, Q @ P[0], atol=1e-5)


def test_als_init_matches_reference_recipe(oracle):
    "(N(0,1) float32 * 0.01)^2, items first then users from ONE generator (_common.py:287-301)."
    rng = np.random.default_rng(7)
    q = oracle.als_initial_params(rng, 5, 3)
    p = oracle.als_initial_params(rng, 4, 3)
    rng2 = np.random.default_rng(7)
    q2 = rng2.standard_normal((5, 3), dtype=np.float32) * 0.01
    q2 *= q2
    p2 = rng2.standard_normal((4, 3), dtype=np.float32) * 0.01
    p2 *= p2
    assert np.array_equal(q, q2) and np.array_equal(p, p2)


def test_als_explicit_row_matches_float64(oracle, rng):
    """explicit.rs:80-119 restated: A = M^T M + reg n I, rhs M^T r; against float64, with an
    empty row (zeros, no delta) and the returned Frobenius delta."""
    import scipy.sparse as sps

    n_rows, n_cols, k = 90, 60, 12
    mask = rng.random((n_rows, n_cols)) < 0.15
    mask[4, :] = False
    vals = rng.standard_normal((n_rows, n_cols)).astype(np.float32)
    m = sps.csr_array(np.where(mask, vals, 0).astype(np.float32))
    m.eliminate_zeros()
    Q = oracle.als_explicit_initial_params(rng, n_cols, k)
    P = oracle.als_explicit_initial_params(rng, n_rows, k)
    assert np.allclose(np.linalg.norm(Q, axis=1), 1.0, atol=1e-6)  # _explicit.py:104-108
    P1 = P.copy()
    frob = oracle.als_explicit_half_epoch(m, P1, Q, 0.25)
    want = oracle.als_explicit_half_epoch_f64(m, Q, 0.25)
    assert np.allclose(P1, want, rtol=1e-4, atol=1e-5)
    assert np.all(P1[4] == 0)
    rows = np.arange(n_rows) != 4  # the empty row is zeroed but contributes no delta
    assert frob == pytest.approx(np.sqrt(((P1 - P)[rows] ** 2).sum()), rel=1e-4)


def test_transpose_csr_matches_scipy_and_reference_loops(oracle, rng):
    """src/accel/data/transpose.rs:42-108 restated with a stable argsort, against the
    reference's three loops written out and SciPy's transpose."""
    import scipy.sparse as sps

    m = sps.random(57, 33, density=0.2, format="csr", dtype=np.float32, random_state=3)
    m.sort_indices()
    ptr, idx, perm = oracle.transpose_csr(m.indptr, m.indices, 33)
    rp = np.zeros(34, dtype=m.indptr.dtype)
    for c in m.indices:
        rp[c + 1] += 1
    rp = np.cumsum(rp).astype(m.indptr.dtype)
    ips = rp.copy()
    ci = np.zeros(m.nnz, np.int32)
    pm = np.zeros(m.nnz, m.indptr.dtype)
    i = 0
    for r in range(57):
        for e in range(m.indptr[r], m.indptr[r + 1]):
            c = m.indices[e]
            ci[ips[c]] = r
            pm[ips[c]] = i
            ips[c] += 1
            i += 1
    assert np.array_equal(ptr, rp) and np.array_equal(idx, ci) and np.array_equal(perm, pm)
    t = sps.csr_array(m.T)
    t.sort_indices()
    assert np.array_equal(ptr, t.indptr) and np.array_equal(idx, t.indices)
    assert np.array_equal(m.data[perm], t.data)


def test_oracle_properties_hypothesis(oracle):
    """Randomised invariants of the integer / selection oracles (they are the checkers of the
    GPU tests, so they get their own adversarial inputs): transpose twice = identity and the
    permutation is a bijection; top-N is a prefix of the full descending sort, never contains
    NaN, and is stable under appending smaller values."""
    import scipy.sparse as sps
    from hypothesis import given, settings
    from hypothesis import strategies as st

    @settings(max_examples=40, deadline=None)
    @given(st.integers(1, 30), st.integers(1, 30), st.floats(0.0, 0.6), st.integers(0, 2**31 - 1))
    def transpose_roundtrip(n_rows, n_cols, density, seed):
        m = sps.random(n_rows, n_cols, density=density, format="csr", dtype=np.float32,
                       random_state=seed)
        m.sort_indices()
        ptr, idx, perm = oracle.transpose_csr(m.indptr, m.indices, n_cols)
        assert sorted(perm.tolist()) == list(range(m.nnz))
        assert ptr[0] == 0 and ptr[-1] == m.nnz and np.all(np.diff(ptr) >= 0)
        ptr2, idx2, perm2 = oracle.transpose_csr(ptr, idx, n_rows)
        assert np.array_equal(ptr2, m.indptr) and np.array_equal(idx2, m.indices)
        assert np.array_equal(perm[perm2], np.arange(m.nnz))

    @settings(max_examples=60, deadline=None)
    @given(st.lists(st.one_of(st.floats(-5, 5, width=32), st.just(float("nan"))), min_size=1,
                    max_size=60), st.integers(1, 70))
    def topn_prefix(values, n):
        s = np.asarray(values, dtype=np.float32)
        top = oracle.argtopn(s, n)
        full = oracle.argsort_descending(s)
        valid = int(np.sum(~np.isnan(s)))
        assert len(top) == min(n, valid) and len(full) == valid
        assert not np.isnan(s[top]).any()
        assert np.array_equal(s[top], s[full][: len(top)])  # same scores, position by position
        assert np.all(np.diff(s[top]) <= 0)
        bigger = np.concatenate([s, np.full(3, -9.0, np.float32)])
        if len(top) == n:
            assert np.array_equal(s[top], bigger[oracle.argtopn(bigger, n)])

    transpose_roundtrip()
    topn_prefix()


def test_user_knn_golden_predictions(oracle, ml_small):
    """The oracle's restatement of `user_score_items_explicit` (src/accel/knn/user_score.rs) +
    `UserKNNScorer` (src/lenskit/knn/user.py) reproduces the reference's golden vector
    tests/models/user-user-preds.csv (UserKNNScorer(k=30, min_sim=1e-6),
    tests/models/test_knn_user_user.py:204-237; the reference's own tolerance is 0.01)."""
    import pandas as pd
    import scipy.sparse as sps

    uv, ur, means = oracle.uknn_prepare(sps.csr_array(ml_small["rmat"]), True)
    known = pd.read_csv(GOLDEN / "user-user-preds.csv")
    assert len(known) == 1756
    worst, missing = 0.0, 0
    for uid, g in known.groupby("user_id"):
        uidx = int(np.searchsorted(ml_small["user_ids"], uid))
        items = np.searchsorted(ml_small["item_ids"], g.item_id.values).astype(np.int32)
        p = oracle.uknn_predict(uv, ur, means.ravel(), uidx, items, 30, 1, 1e-6, True)
        missing += int(np.sum(np.isnan(p) & ~np.isnan(g.prediction.values)))
        err = np.abs(p - g.prediction.values)
        worst = max(worst, float(err[~np.isnan(err)].max()))
    assert missing == 0 and worst < 1e-4, (missing, worst)


def test_ease_oracle_closed_form(oracle, rng):
    "oracle.ease_train against the textbook EASE solution B = I - P diag(1/diag P), P = (G + reg I)^-1"
    import scipy.sparse as sps

    m = sps.random(80, 30, density=0.15, random_state=rng, format="csr", dtype=np.float32)
    m.data[:] = 1.0
    w = oracle.ease_train(sps.csr_array(m), 3.0)
    x = m.toarray().astype(np.float64)
    p = np.linalg.inv(x.T @ x + 3.0 * np.eye(30))
    b = np.eye(30) - p / np.diag(p).reshape(1, -1)
    b[np.diag_indices(30)] = 0.0
    assert np.all(np.diag(w) == 0.0)
    assert np.allclose(w, b, rtol=1e-4, atol=1e-6)
    hist = np.array([1, 4, 9])
    assert np.allclose(oracle.ease_score(w, hist), w[hist].sum(axis=0), rtol=1e-6, atol=1e-7)


def test_parity_accounting_attributes_gaps():
    """
    oracle/parity.py: a decidable row over 1e-4 vs the oracle counts against the GPU only when the
    GPU row itself is off the float64 answer; a drift of the reference arithmetic is listed.
    """
    from oracle import parity

    rng = np.random.default_rng(0)
    exact = rng.standard_normal((100, 8))
    got = (exact * (1 + 1e-6)).astype(np.float32)
    want = exact.astype(np.float32).copy()
    want[3] *= 1 + 2e-4  # the reference drifts on row 3
    cond = np.full(100, 80.0)
    a = parity.als_half_accounting(got, want, exact, cond)
    assert a["rows_over_1e-4"] == 1 and a["decidable_rows_over_1e-4"] == 1
    assert a["decidable_rows_over_1e-4_gpu_side"] == 0 and a["accounted"]
    assert not a["ok"]  # the raw criterion (no row over 1e-4) is reported as it is
    assert a["exceptions"][0]["row"] == 3
    got2 = got.copy()
    got2[5] *= 1 + 2e-4  # now the GPU is the one that is off
    b = parity.als_half_accounting(got2, exact.astype(np.float32), exact, cond)
    assert b["decidable_rows_over_1e-4_gpu_side"] == 1 and not b["accounted"] and not b["ok"]
    # without a referee the old, stricter rule applies
    c = parity.als_half_accounting(got, want, None, cond)
    assert not c["accounted"] and not c["ok"]
    d = parity.als_half_accounting(got, exact.astype(np.float32), exact, cond)
    assert d["ok"] and d["accounted"] and d["rows_over_1e-4"] == 0


def test_batch_query_entry_points_equal_the_per_query_functions(oracle, rng):
    "lko_score_topn_batch / lko_iknn_score_batch = the per-query restatements, query by query"
    Q = rng.standard_normal((3000, 24)).astype(np.float32)
    P = rng.standard_normal((40, 24)).astype(np.float32)
    lens = rng.integers(0, 30, 40)
    ptr = np.zeros(41, np.int64)
    ptr[1:] = np.cumsum(lens)
    ex = rng.integers(0, 3000, ptr[-1]).astype(np.int32)
    gi, gs = oracle.score_topn_batch(Q, P, 50, ptr, ex)
    for b in range(40):
        sc = oracle.score_dense(Q, P[b])
        sc[ex[ptr[b]:ptr[b + 1]]] = np.nan
        w = oracle.argtopn(sc, 50)
        assert np.array_equal(gi[b], w) and np.array_equal(gs[b], sc[w])
    # a catalogue smaller than n: padded with -1 / NaN
    gi2, gs2 = oracle.score_topn_batch(Q[:20], P[:3], 50)
    assert (gi2[:, 20:] == -1).all() and np.isnan(gs2[:, 20:]).all() and (gi2[:, :20] >= 0).all()

    sims = sps.random_array((200, 200), density=0.2, format="csr", dtype=np.float32,
                            rng=np.random.default_rng(1))
    sims.sort_indices()
    rl = rng.integers(0, 25, 30)
    rp = np.zeros(31, np.int64)
    rp[1:] = np.cumsum(rl)
    ri = rng.integers(0, 200, rp[-1]).astype(np.int32)
    rr = rng.standard_normal(rp[-1]).astype(np.float32)
    tp = np.arange(31, dtype=np.int64) * 12
    ti = rng.integers(0, 200, 30 * 12).astype(np.int32)
    bs, bc = oracle.iknn_score_batch(sims, rp, ri, rr, tp, ti, 10, 2)
    for q in range(30):
        s1, c1 = oracle.iknn_score(sims, ri[rp[q]:rp[q + 1]], rr[rp[q]:rp[q + 1]],
                                   ti[tp[q]:tp[q + 1]], 10, 2)
        assert np.array_equal(bc[tp[q]:tp[q + 1]], c1)
        assert np.array_equal(np.isnan(bs[tp[q]:tp[q + 1]]), np.isnan(s1))
        ok = ~np.isnan(s1)
        assert np.array_equal(bs[tp[q]:tp[q + 1]][ok], s1[ok])


def test_iknn_recommend_batch_is_the_per_query_pipeline(oracle):
    """The recommend restatement = per query: ``iknn_score`` over the candidates (all items minus
    the query's own), means added back, ``argtopn`` -- the pinned pieces, composed
    (src/lenskit/knn/item.py:231-295, basic/candidates.py:77-94, basic/topn.py:45-69)."""
    import scipy.sparse as sps

    rng = np.random.default_rng(5)
    n_users, n_items = 80, 300
    lens = rng.integers(3, 40, n_users)
    rows = np.repeat(np.arange(n_users), lens)
    cols = np.concatenate([rng.choice(n_items, ln, replace=False) for ln in lens])
    rmat = sps.coo_array((rng.integers(1, 11, len(rows)).astype(np.float32) * 0.5, (rows, cols)),
                         shape=(n_users, n_items))
    ui, iu, means, _ = oracle.iknn_prepare(rmat, True)
    sims = oracle.iknn_build(ui, iu, 1.0e-6, None)
    means = np.asarray(means, np.float32).ravel()
    csr = sps.csr_array(rmat)
    users = [0, 5, 9]
    ptr = np.concatenate([[0], np.cumsum(np.diff(csr.indptr)[users])]).astype(np.int64)
    idx = np.concatenate([csr.indices[csr.indptr[u]:csr.indptr[u + 1]] for u in users]).astype(np.int32)
    val = np.concatenate([csr.data[csr.indptr[u]:csr.indptr[u + 1]] for u in users]) - means[idx]
    wi, ws, full = oracle.iknn_recommend_batch(sims, ptr, idx, val.astype(np.float32), means,
                                               10, 2, 7)
    assert wi.shape == (3, 7) and full.shape == (3, n_items)
    for q, u in enumerate(users):
        own = idx[ptr[q]:ptr[q + 1]]
        cand = np.setdiff1d(np.arange(n_items, dtype=np.int32), own)
        sc, _cnt = oracle.iknn_score(sims, own, val[ptr[q]:ptr[q + 1]].astype(np.float32), cand,
                                     10, 2)
        sc = sc + means[cand]
        top = oracle.argtopn(sc, 7)
        assert np.array_equal(ws[q][: len(top)].view(np.uint32), sc[top].view(np.uint32))
        assert np.array_equal(wi[q][: len(top)], cand[top])
        assert not np.isin(wi[q][wi[q] >= 0], own).any()
