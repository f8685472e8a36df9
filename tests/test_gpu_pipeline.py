"""GPU: the reference's pipeline TOMLs end to end through the component API."""
import pickle
from pathlib import Path

import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).parent / "golden"


@pytest.fixture(scope="module")
def ml_ds():
    from lkpy_amd.data import load_movielens_npz

    return load_movielens_npz(GOLDEN / "ml_small.npz")


def _rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def test_als_implicit_toml_trains_like_the_reference(gpu, oracle, ml_small, ml_ds):
    """pipelines/als-implicit.toml, seed 42: same seed handling and init as the reference
    => the scorer sees SeedSequence(42).spawn child (2,), items initialised first."""
    from lkpy_amd.pipeline import Pipeline
    from lkpy_amd.training import TrainingOptions

    pipe = Pipeline.load_config(GOLDEN / "pipelines" / "als-implicit.toml")
    scorer = pipe.node("scorer").component
    scorer.config.epochs = 5
    pipe.train(ml_ds, TrainingOptions(rng=42))
    assert scorer.is_trained() and scorer.trained_epochs == 5
    assert scorer.user_embeddings.shape == (671, 64) and scorer.item_embeddings.shape == (9125, 64)
    assert scorer._OtOr.shape == (64, 64)

    rmat = ml_small["rmat"]
    ind = sps.coo_array((np.ones(rmat.nnz, np.float32), (rmat.row, rmat.col)), rmat.shape)
    want = oracle.als_train(ind, 64, 5, np.random.SeedSequence(42).spawn(3)[2])
    # two float32 implementations of an ill-conditioned iteration: ~1e-3 apart (see
    # test_gpu_als.py::test_half_epoch_ml_small_cfg1 for the per-half-epoch analysis)
    traj = (_rel(scorer.item_embeddings, want.item_embeddings),
            _rel(scorer.user_embeddings, want.user_embeddings), _rel(scorer._OtOr, want.OtOr))
    print("\n5-epoch trajectories GPU vs oracle (Q, P, OtOr):", ["%.2e" % t for t in traj])
    assert max(traj) < 4e-3  # (measured 1.25e-3 / 9.8e-4 / 1.0e-3; rounds 1-5 allowed 2e-2)
    empty = np.bincount(ind.col, minlength=9125) == 0
    assert np.all(scorer.item_embeddings[empty] == 0)

    # The 5-epoch trajectories above are two float32 iterations of an ill-conditioned map; the
    # per-row statement is one more epoch FROM IDENTICAL INPUTS (the trained factors): every row
    # whose system lets two float32 solves agree (cond u < 2.5e-5) within the raw 1e-4 of the
    # oracle's row on the default path, the others counted
    from lkpy_amd._als_engine import HipBackend, ImplicitALSEngine

    ui6 = sps.csr_array(oracle.als_prepare_matrix(ind, 40.0))
    ui6.sort_indices()
    iu6 = sps.csr_array(ui6.T)
    iu6.sort_indices()
    P5, Q5 = scorer.user_embeddings.copy(), scorer.item_embeddings.copy()
    ureg = ireg = 0.1
    eng = ImplicitALSEngine(ui6, 64, ureg, ireg, P5, Q5, HipBackend(64, gpu))
    eng.train_epoch()
    eng.check()
    P6, Q6 = eng.user_embeddings(), eng.item_embeddings()
    for name, mat, this, other, got, reg_ in (("user", ui6, P5, Q5, P6, ureg),
                                              ("item", iu6, Q5, P6, Q6, ireg)):
        w = np.ascontiguousarray(this.copy())
        oracle.als_half_epoch(mat, w, other, oracle.implicit_otor(other, reg_))
        _x, cond = oracle.als_referee_f64(mat, other, reg_)
        den = np.linalg.norm(w.astype(np.float64), axis=1)
        nz = den > 0
        e = np.linalg.norm(got.astype(np.float64) - w, axis=1)[nz] / den[nz]
        decid = cond[nz] * 2.0**-24 < 2.5e-5
        mx = lambda a: float(a.max()) if a.size else 0.0  # noqa: E731
        assert (e[decid] <= 1e-4).all(), (name, mx(e[decid]))
        # (on this data set cond(A) > 1e3 for nearly every row: the decidable set may be empty;
        # every row must still sit inside the forward bound of a float32 solve RELATIVE TO THE
        # ORACLE: two backward-stable solves of one system differ by at most ~2 x 16 cond u)
        cu = cond[nz] * 2.0**-24
        assert (e <= 32.0 * cu + 2e-6).all(), (name, float((e / cu).max()))
        # the undecidable rows, pinned (a regression must not hide there): measured on an
        # MI355X in round 6 -- user half 9 rows over 1e-4 (worst 4.1e-4), item half 1064 (2.6e-4)
        cap_n, cap_e = {"user": (20, 8.0e-4), "item": (1600, 5.2e-4)}[name]
        assert int((e[~decid] > 1e-4).sum()) <= cap_n and mx(e[~decid]) <= cap_e, \
            (name, int((e[~decid] > 1e-4).sum()), mx(e[~decid]))
        print(f"\ncfg1 epoch 6 from identical inputs, {name} half: {int(decid.sum())} rows with "
              f"cond u < 2.5e-5, all within 1e-4 (max {mx(e[decid]):.1e}); other rows "
              f"{int((~decid).sum())}, of those over 1e-4: {int((e[~decid] > 1e-4).sum())} (max "
              f"{mx(e[~decid]):.1e}; max err / (cond u) {float((e / cu).max()):.2f})")

    # recommend through the pipeline == the reference's steps restated on the CPU from the
    # SAME factors: history lookup, candidates minus history, fold-in, scores, top-N
    csr = sps.csr_array(ind)
    Q, OtOr = scorer.item_embeddings, scorer._OtOr
    from lkpy_amd.data import ItemList as _IL

    users = ml_small["user_ids"][::29]
    same_cpu_foldin, near_ties, fold_err = 0, 0, 0.0
    for uid in users:
        recs = pipe.run("recommender", query=int(uid), n=10)
        assert len(recs) == 10 and recs.ordered
        u = int(np.searchsorted(ml_small["user_ids"], uid))
        hist = csr.indices[csr.indptr[u] : csr.indptr[u + 1]]
        got = recs.numbers(vocabulary=scorer.items)
        assert not np.isin(got, hist).any()
        # (1) index sets BIT-EXACT: from the SAME query vector (the GPU's fold-in) the reference's
        # scoring + candidate exclusion + heap top-N give exactly the GPU's list and score bits
        x_gpu, _ = scorer.new_user_embedding(None, _IL(item_nums=hist, vocabulary=scorer.items))
        s = oracle.score_dense(Q, x_gpu.astype(np.float32))
        s[hist] = np.nan
        top = oracle.argtopn(s, 10)
        assert np.array_equal(got, top), (int(uid), got, top)
        assert np.array_equal(np.asarray(recs.scores(), np.float32).view(np.uint32),
                              s[top].view(np.uint32))
        # (2) the fold-in vector itself vs the reference's _train_new_row restatement: this
        # system is ill-conditioned on ml-latest-small (cond 1e3..2e5, test_gpu_als_reference.py)
        x_cpu = oracle.als_fold_in(hist, np.full(len(hist), 40.0, np.float32), Q, OtOr)
        fold_err = max(fold_err, _rel(x_gpu, x_cpu))
        # (3) with the CPU fold-in vector instead, lists may differ only at NEAR-TIES: wherever
        # they differ, the two scores involved are closer than the fold-in difference explains
        s2 = oracle.score_dense(Q, x_cpu.astype(np.float32))
        s2[hist] = np.nan
        top2 = oracle.argtopn(s2, 10)
        if np.array_equal(got, top2):
            same_cpu_foldin += 1
        else:
            d = got != top2
            gap = np.abs(s2[got[d]] - s2[top2[d]]) / np.maximum(np.abs(s2[top2[d]]), 1e-12)
            assert gap.max() < 1e-3, (int(uid), gap)
            near_ties += 1
    print(f"\nrecommend parity over {len(users)} users: lists bit-identical given the same query "
          f"vector: {len(users)}/{len(users)}; with the CPU fold-in vector: {same_cpu_foldin} "
          f"identical, {near_ties} differ at near-ties only; fold-in rel err max {fold_err:.2e}")
    assert fold_err < 3e-4  # (measured 7.3e-5 with fold-in plans in the kernels' own order -- the
    # reference's fold-in is NumPy, not the Rust chain; 3.9e-4 in the hybrid order; round 5 allowed 5e-3)

    # NaN semantics (tests/models/test_als_implicit.py:277-298, _common.py:145-170)
    from lkpy_amd.data import ItemList

    res = scorer(query=int(users[0]), items=ItemList([1, 2, -999]))
    assert np.isnan(res.scores()[2]) and res.scores().dtype == np.float32
    assert np.all(np.isnan(scorer(query=-12345, items=ItemList([1, 2])).scores()))

    # pickle round trip: scores equal within 1e-3 (testing/_components.py:146-188)
    clone = pickle.loads(pickle.dumps(pipe))
    r1 = pipe.run("scorer", query=int(users[1]), items=ItemList(ml_ds.items.ids()[:500]))
    r2 = clone.run("scorer", query=int(users[1]), items=ItemList(ml_ds.items.ids()[:500]))
    assert np.allclose(r1.scores(), r2.scores(), atol=1e-3, equal_nan=True)


def test_als_batch_recommend_equals_per_query(gpu, ml_ds):
    from lkpy_amd import batch
    from lkpy_amd.als import ImplicitMFScorer
    from lkpy_amd.pipeline import topn_pipeline
    from lkpy_amd.training import TrainingOptions

    pipe = topn_pipeline(ImplicitMFScorer(features=25, epochs=3))
    pipe.train(ml_ds, TrainingOptions(rng=7))
    users = [int(u) for u in ml_ds.users.ids()[::40]] + [-1]
    out = batch.recommend(pipe, users, 20)
    # (an ItemListCollection keyed by user_id: src/lenskit/batch/_runner.py:157-191)
    assert len(out.lookup(-1)) == 0  # unknown user without history: nothing to recommend
    assert [k.user_id for k in out.keys()] == users and out[0][0].user_id == users[0]
    for u in users[:-1]:
        one = pipe.run("recommender", query=u, n=20)
        assert np.array_equal(out.lookup(u).ids(), one.ids())
        assert np.array_equal(out.lookup(user_id=u).scores(), one.scores())


def test_user_embeddings_false_and_prefer(gpu, ml_ds):
    "tests/models/test_als_implicit.py:301-324 and the 'prefer' path of _common.py:145-157."
    from lkpy_amd.als import ImplicitMFScorer
    from lkpy_amd.data import ItemList
    from lkpy_amd.pipeline import topn_pipeline
    from lkpy_amd.training import TrainingOptions

    pipe = topn_pipeline(ImplicitMFScorer(features=16, epochs=2, user_embeddings=False))
    pipe.train(ml_ds, TrainingOptions(rng=1))
    sc = pipe.node("scorer").component
    assert sc.user_embeddings is None and sc.users is None
    assert len(pipe.run("recommender", query=int(ml_ds.users.id(3)), n=5)) == 5
    pipe = topn_pipeline(ImplicitMFScorer(features=16, epochs=2, user_embeddings="prefer"))
    pipe.train(ml_ds, TrainingOptions(rng=1))
    sc = pipe.node("scorer").component
    uid = int(ml_ds.users.id(3))
    items = ItemList(ml_ds.items.ids()[:64])
    got = sc(query=pipe.node("history-lookup").component(uid), items=items).scores()
    want = sc.item_embeddings[:64] @ sc.user_embeddings[3]
    assert np.allclose(got, want, rtol=1e-4, atol=1e-6)


def test_iknn_explicit_toml_end_to_end(gpu, oracle, ml_small, ml_ds):
    from lkpy_amd import batch
    from lkpy_amd.data import ItemList
    from lkpy_amd.pipeline import Pipeline

    pipe = Pipeline.load_config(GOLDEN / "pipelines" / "iknn-explicit.toml")
    pipe.train(ml_ds)
    knn = pipe.node("scorer").component
    ui, iu, means, _ = oracle.iknn_prepare(ml_small["rmat"], True)
    want = oracle.iknn_build(ui, iu, 1.0e-6, None)
    import pyarrow as pa

    sm = knn.sim_matrix  # Arrow extension array, LargeList storage (knn/item.py:176-177)
    assert pa.types.is_large_list(sm.type.storage_type) and sm.shape == want.shape
    assert np.array_equal(sm.offsets.to_numpy(), want.indptr)
    assert np.array_equal(sm.indices.to_numpy(), want.indices)
    assert np.array_equal(sm.values.to_numpy().view(np.uint32), want.data.view(np.uint32))
    assert np.array_equal(knn.item_means, means)
    assert np.array_equal(knn.item_counts, np.diff(want.indptr))

    # the reference's golden predictions through batch.predict (k=20 is the TOML default)
    known = pd.read_csv(GOLDEN / "item-item-preds.csv")
    pairs = {int(u): ItemList(g.item_id.values) for u, g in known.groupby("user_id")}
    preds = batch.predict(pipe, pairs)
    got = np.concatenate([preds.lookup(int(u)).scores() for u, _ in known.groupby("user_id")])
    exp = np.concatenate([g.prediction.values for _, g in known.groupby("user_id")])
    assert not np.any(np.isnan(got) & ~np.isnan(exp))
    err = np.abs(got - exp)
    assert np.sum(err > 1e-5) <= 5 and np.median(err) < 1e-6

    # per-query path, unknown items, fallback merge (std:topn-predict)
    uid = int(known.user_id.iloc[0])
    res = pipe.run("scorer", query=uid, items=ItemList([int(known.item_id.iloc[0]), -7]))
    assert np.isfinite(res.scores()[0]) and np.isnan(res.scores()[1])
    assert res.field("nbr_counts")[0] > 0
    full = pipe.run("rating-predictor", query=uid, items=ItemList(ml_ds.items.ids()[:200]))
    assert np.all(np.isfinite(full.scores()))  # the bias fallback fills the gaps
    recs = pipe.run("recommender", query=uid, n=10)
    assert len(recs) == 10 and np.all(np.diff(recs.scores()) <= 0)
    hist = ml_ds.user_row(uid).ids()
    assert not np.isin(recs.ids(), hist).any()
    # no history => all NaN (item.py:238-245)
    assert np.all(np.isnan(knn(query=-5, items=ItemList([1, 2, 3])).scores()))
    clone = pickle.loads(pickle.dumps(knn))
    assert clone.sim_matrix.equals(knn.sim_matrix)
    r2 = clone(query=pipe.node("history-lookup").component(uid), items=ItemList(ml_ds.items.ids()[:50]))
    r1 = knn(query=pipe.node("history-lookup").component(uid), items=ItemList(ml_ds.items.ids()[:50]))
    assert np.array_equal(r1.scores(), r2.scores(), equal_nan=True)


def test_iknn_constant_ratings_warns(gpu):
    from lkpy_amd.data import from_interactions_df
    from lkpy_amd.knn import DataWarning, ItemKNNScorer

    df = pd.DataFrame({"user_id": [1, 1, 2, 2, 3, 3], "item_id": [1, 2, 1, 2, 1, 2], "rating": 1.0})
    with pytest.warns(DataWarning):
        ItemKNNScorer(k=5).train(from_interactions_df(df))


def test_function_seam(gpu, oracle, ml_small, rng):
    "The `_accel` stand-ins with the reference's argument lists, through run_accel_task."
    from lkpy_amd import _accel
    from lkpy_amd.data import SparseRowArray
    from lkpy_amd.parallel import run_accel_task

    rmat = ml_small["rmat"]
    ind = sps.coo_array((np.full(rmat.nnz, 40.0, np.float32), (rmat.row, rmat.col)), rmat.shape)
    ui = sps.csr_array(ind)
    k = 25
    other = (rng.standard_normal((ui.shape[1], k)) * 0.1).astype(np.float32)
    this = (rng.standard_normal((ui.shape[0], k)) * 0.1).astype(np.float32)
    otor = oracle.implicit_otor(other, 0.1)
    want = this.copy()
    wd = oracle.als_half_epoch(ui, want, other, otor)
    d = run_accel_task(_accel.als.train_implicit_matrix(SparseRowArray.from_scipy(ui), this,
                                                        other, otor))
    assert d == pytest.approx(wd, rel=1e-4)
    assert _rel(this, want) < 1e-4  # updated IN PLACE
    with pytest.raises(TypeError):
        _accel.als.train_implicit_matrix(ui, this.astype(np.float64), other, otor)

    uin, iun, _m, _ = oracle.iknn_prepare(ml_small["rmat"], True)
    chunks = run_accel_task(_accel.knn.compute_similarities(
        SparseRowArray.from_scipy(uin), SparseRowArray.from_scipy(iun), uin.shape, 1e-6, None))
    import pyarrow as pa

    assert isinstance(chunks, list) and isinstance(chunks[0], pa.LargeListArray)
    ws = oracle.iknn_build(uin, iun, 1e-6, None)
    got_vals = chunks[0].values.field("value").to_numpy()
    assert np.array_equal(got_vals.view(np.uint32), ws.data.view(np.uint32))

    s = rng.standard_normal(5000).astype(np.float32)
    s[::7] = np.nan
    assert np.array_equal(_accel.data.argtopn(s, 50), oracle.argtopn(s, 50))
    assert len(_accel.data.argtopn(s, 0)) == 0
    full = _accel.data.argsort_descending(s[:3000])
    assert np.array_equal(full, oracle.argsort_descending(s[:3000]))
    import pyarrow as pa

    assert np.array_equal(_accel.data.argtopn(pa.array(s[:100]), 5), oracle.argtopn(s[:100], 5))


def test_als_explicit_toml_trains_like_the_reference(gpu, oracle, ml_small, ml_ds):
    """
    ``pipelines/als-explicit.toml`` (the reference's file) through the component seam
    (SURVEY.md 8f-2): bias model on the host, ALS epochs in the explicit mode of the HIP
    kernel.  The trained factors are compared with the oracle run from the same init on
    the same normalised matrix; predictions stay in the rating range, fold-in agrees with
    the stored embedding, unknown items/users give NaN
    (tests/models/test_als_explicit.py:48-180).
    """
    from lkpy_amd import als
    from lkpy_amd.data import ItemList, RecQuery
    from lkpy_amd.pipeline import Pipeline
    from lkpy_amd.training import Trainable, TrainingOptions

    pipe = Pipeline.load_config(GOLDEN / "pipelines" / "als-explicit.toml")
    scorer = pipe.node("scorer").component
    assert isinstance(scorer, als.BiasedMFScorer)
    scorer.config.embedding_size = 20
    scorer.config.epochs = 4
    pipe.train(ml_ds, TrainingOptions(rng=42))
    assert scorer.is_trained() and scorer.trained_epochs == 4
    assert scorer.user_embeddings.shape == (671, 20) and scorer.item_embeddings.shape == (9125, 20)
    rmat = ml_small["rmat"]
    assert scorer.bias.global_bias == pytest.approx(float(np.mean(rmat.data)))

    # oracle: same bias model, same seed child / init recipe, same epoch order
    norm = scorer.bias.transform_matrix(sps.coo_array(rmat)).astype(np.float32)
    ui = sps.csr_array(norm)
    iu = sps.csr_array(ui.T)
    ui.sort_indices()
    iu.sort_indices()
    trainables = [n for n, node in pipe.nodes.items()
                  if node.component is not None and isinstance(node.component, Trainable)]
    idx = trainables.index("scorer")
    rng = np.random.default_rng(np.random.SeedSequence(42).spawn(idx + 1)[idx])
    Q = oracle.als_explicit_initial_params(rng, ui.shape[1], 20)
    P = oracle.als_explicit_initial_params(rng, ui.shape[0], 20)
    for _ in range(4):
        oracle.als_explicit_half_epoch(ui, P, Q, scorer.config.user_reg)
        oracle.als_explicit_half_epoch(iu, Q, P, scorer.config.item_reg)
    assert _rel(scorer.user_embeddings, P) < 5e-3 and _rel(scorer.item_embeddings, Q) < 5e-3
    empty = np.bincount(rmat.col, minlength=9125) == 0
    assert np.all(scorer.item_embeddings[empty] == 0)

    # predictions: rating range, fold-in close to the stored row, NaN semantics
    uid = int(ml_small["user_ids"][10])
    items = ItemList(ml_ds.items.ids()[:200])
    stored = pipe.run("scorer", query=uid, items=items).scores()
    assert np.all(np.isfinite(stored)) and stored.min() > -1.0 and stored.max() < 7.0
    hist = ml_ds.user_row(uid)
    folded = scorer(RecQuery(user_id=uid, user_items=hist), items).scores()
    assert np.allclose(folded, stored, rtol=9e-2, atol=0.3)
    res = scorer(query=uid, items=ItemList([1, 2, -999]))
    assert np.isnan(res.scores()[2]) and np.all(np.isfinite(res.scores()[:2]))
    assert np.all(np.isnan(scorer(query=-12345, items=ItemList([1, 2])).scores()))
    # rating-predictor of the std:topn-predict pipeline
    pred = pipe.run("rating-predictor", query=uid, items=ItemList([1, 2, 3]))
    assert np.all(np.isfinite(pred.scores()))
    recs = pipe.run("recommender", query=uid, n=10)
    assert len(recs) == 10 and recs.ordered
    clone = pickle.loads(pickle.dumps(pipe))
    r2 = clone.run("scorer", query=uid, items=items)
    assert np.allclose(stored, r2.scores(), atol=1e-3)


def test_repeated_pairs_are_summed_like_the_reference(gpu):
    """``prepare_matrix`` sums repeated (user, item) pairs as the reference's
    COO -> CSR conversion does; a dataset with a pair split in two trains to the same factors
    as the dataset holding the summed rating once."""
    from lkpy_amd.als import ImplicitMFScorer
    from lkpy_amd.data import Dataset
    from lkpy_amd.training import TrainingOptions

    rng = np.random.default_rng(5)
    u = rng.integers(0, 200, 6000)
    i = rng.integers(0, 300, 6000)
    key = np.unique(u * 1000 + i)
    u, i = key // 1000, key % 1000
    r = rng.integers(1, 6, len(u)).astype(np.float32)
    # split the first 500 pairs into two interactions carrying half the rating each
    u2, i2 = np.concatenate([u, u[:500]]), np.concatenate([i, i[:500]])
    r2 = np.concatenate([r, r[:500] / 2])
    r2[:500] /= 2
    all_items = np.arange(300)
    a = ImplicitMFScorer(features=16, epochs=3, use_ratings=True)
    a.train(Dataset.from_arrays(u, i, r, all_item_ids=all_items), TrainingOptions(rng=7))
    b = ImplicitMFScorer(features=16, epochs=3, use_ratings=True)
    dsb = Dataset.from_arrays(u2, i2, r2, all_item_ids=all_items)
    assert dsb.has_duplicates
    b.train(dsb, TrainingOptions(rng=7))
    assert np.array_equal(a.item_embeddings, b.item_embeddings)
    assert np.array_equal(a.user_embeddings, b.user_embeddings)
