"""CPU: host-side mirror of the reference interface (no kernels involved)."""
import pickle
from pathlib import Path

import numpy as np
import pandas as pd
import pytest

GOLDEN = Path(__file__).parent / "golden"


def test_reference_tomls_load_unchanged():
    """tests/pipeline/test_human_config.py:24-39 -- als-implicit.toml yields an
    ImplicitMFScorer scorer and a TopNRanker recommender; iknn-explicit.toml (std:topn-predict)
    adds the fallback predictor."""
    from lkpy_amd.als import ImplicitMFScorer
    from lkpy_amd.basic import BiasScorer, FallbackScorer, TopNRanker
    from lkpy_amd.knn import ItemKNNScorer
    from lkpy_amd.pipeline import Pipeline

    p = Pipeline.load_config(GOLDEN / "pipelines" / "als-implicit.toml")
    assert isinstance(p.node("scorer").component, ImplicitMFScorer)
    assert p.node("scorer").component.config.user_embeddings is True
    assert isinstance(p.node("recommender").component, TopNRanker)
    assert list(p.nodes)[:3] == ["query", "items", "n"]
    p = Pipeline.load_config(GOLDEN / "pipelines" / "iknn-explicit.toml")
    assert isinstance(p.node("scorer").component, ItemKNNScorer)
    assert isinstance(p.node("fallback-predictor").component, BiasScorer)
    assert isinstance(p.node("rating-predictor").component, FallbackScorer)
    from lkpy_amd.als import BiasedMFScorer

    p = Pipeline.load_config(GOLDEN / "pipelines" / "als-explicit.toml")
    sc = p.node("scorer").component
    assert isinstance(sc, BiasedMFScorer)
    assert sc.config.damping == 5.0 and sc.config.epochs == 10  # _explicit.py:25-29


def test_config_aliases_and_validation():
    "tests/models/test_als_implicit.py:42-69, test_knn_item_item.py:98-103"
    from lkpy_amd.als import ImplicitMFConfig, ImplicitMFScorer
    from lkpy_amd.knn import ItemKNNScorer

    assert ImplicitMFScorer(features=25).config.embedding_size == 25
    assert ImplicitMFScorer(embedding_size_exp=5).config.embedding_size == 32
    cfg = ImplicitMFConfig(regularization={"user": 0.2, "item": 0.05})
    assert cfg.user_reg == 0.2 and cfg.item_reg == 0.05
    assert ImplicitMFConfig().weight == 40 and ImplicitMFConfig().epochs == 10
    m = ItemKNNScorer(k=30)
    assert m.dump_config()["max_nbrs"] == 30 and m.dump_config()["feedback"] == "explicit"
    assert ItemKNNScorer(min_sim=1e-320).config.min_sim >= np.finfo(np.float64).smallest_normal
    with pytest.raises(Exception):
        ItemKNNScorer(bogus=1)
    with pytest.raises(Exception):
        ImplicitMFScorer(epochs=0)
    assert pickle.loads(pickle.dumps(ImplicitMFScorer(features=8))).config.embedding_size == 8


def test_dataset_vocabulary_itemlist():
    from lkpy_amd.data import ItemList, RecQuery, Vocabulary, load_movielens_npz

    ds = load_movielens_npz(GOLDEN / "ml_small.npz")
    assert (ds.user_count, ds.item_count, ds.interaction_count) == (671, 9125, 100004)
    m = ds.interactions().matrix().scipy(layout="csr")
    assert m.dtype == np.float32 and np.all(m.data == 1.0)  # _relationships.py:603-657
    r = ds.interactions().matrix().scipy("rating", layout="coo")
    assert r.shape == (671, 9125) and set(np.unique(r.data)) <= set(np.arange(1, 11) * 0.5)
    assert np.all(np.diff(ds.items.ids()) > 0)
    v = ds.items
    assert v.number(v.id(17)) == 17 and v.number(-5, missing=None) is None
    with pytest.raises(KeyError):
        v.number(-5)
    assert v.numbers([v.id(3), -1], missing="negative").tolist() == [3, -1]
    row = ds.user_row(ds.users.id(0))
    assert len(row) == m.indptr[1] and row.field("rating") is not None
    il = ItemList([10, 20, 30], scores=[0.5, np.nan, 2.0], vocabulary=Vocabulary([10, 20, 30]))
    assert il.scores().dtype == np.float32 and len(il.remove(numbers=[1])) == 2
    assert ItemList(il, scores=np.nan).scores().tolist() != il.scores().tolist()
    q = RecQuery.create(5)
    assert q.user_id == 5 and q.query_items is None
    assert RecQuery.create(il).query_items is il and RecQuery.create(q) is q


def test_history_candidates_bias_components():
    from lkpy_amd.basic import (BiasScorer, FallbackScorer, TrainingItemsCandidateSelector,
                                UserTrainingHistoryLookup)
    from lkpy_amd.data import ItemList, from_interactions_df

    df = pd.DataFrame({"user_id": [1, 1, 2, 2, 3], "item_id": [10, 20, 10, 30, 20],
                       "rating": [4.0, 3.0, 2.0, 5.0, 1.0]})
    ds = from_interactions_df(df)
    look = UserTrainingHistoryLookup()
    look.train(ds)
    q = look(1)
    assert q.history_items.ids().tolist() == [10, 20]
    assert look(99).history_items is None
    cand = TrainingItemsCandidateSelector()
    cand.train(ds)
    assert cand(q).ids().tolist() == [30]  # training items minus the query's items
    bias = BiasScorer()
    bias.train(ds)
    mu = df.rating.mean()
    assert bias.global_bias == pytest.approx(mu)
    bi = df.assign(c=df.rating - mu).groupby("item_id").c.mean()
    assert bias.item_biases == pytest.approx(bi.values)
    s = bias(q, ItemList([10, 20, 30, 99]))
    assert len(s) == 4 and np.all(np.isfinite(s.scores()))
    fb = FallbackScorer()
    merged = fb(primary=ItemList([10, 20], scores=[np.nan, 2.0]),
                backup=ItemList([10, 20], scores=[1.5, 9.0]))
    assert merged.scores().tolist() == [1.5, 2.0]


def test_pipeline_seed_spawning_matches_reference():
    """Pipeline.train hands the i-th Trainable node the i-th spawned SeedSequence child
    (src/lenskit/pipeline/_impl.py:346-366): the scorer of std:topn gets spawn_key (2,)."""
    from lkpy_amd.pipeline import Component, Pipeline
    from lkpy_amd.training import TrainingOptions

    seen = {}

    class Probe(Component):
        def is_trained(self):
            return False

        def train(self, data, options):
            seen["rng"] = options.rng

        def __call__(self, query, items):
            return items

    p = Pipeline.std_topn()
    p.replace_component("scorer", Probe())
    from lkpy_amd.data import from_interactions_df

    ds = from_interactions_df(pd.DataFrame({"user_id": [1], "item_id": [2], "rating": [3.0]}))
    p.train(ds, TrainingOptions(rng=42))
    assert seen["rng"].spawn_key == (2,) and seen["rng"].entropy == 42


def test_accel_task_protocol():
    from lkpy_amd.parallel import AccelTask, run_accel_task

    t = AccelTask(lambda task: (task.set_progress(7), 42)[1], total=7)
    assert run_accel_task(t) == 42 and t.current_progress() == (7, 7)
    with pytest.raises(RuntimeError, match="accelerator task failed"):
        run_accel_task(AccelTask(lambda task: 1 / 0))
    with pytest.raises(RuntimeError):
        t.invoke()  # invoke exactly once


def test_bias_model_normalisation_round_trip():
    """``BiasModel`` (src/lenskit/basic/bias.py:35-275): damped means in closed form, the
    matrix transform the biased-MF trainer applies, and its inverse at scoring time."""
    import scipy.sparse as sps

    from lkpy_amd.basic import BiasModel
    from lkpy_amd.data import ItemList, from_interactions_df

    df = pd.DataFrame({"user_id": [1, 1, 2, 2, 3, 3], "item_id": [10, 20, 10, 30, 20, 30],
                       "rating": [4.0, 3.0, 2.0, 5.0, 1.0, 4.5]})
    ds = from_interactions_df(df)
    m = BiasModel.learn(ds, damping=5.0)
    mu = df.rating.mean()
    assert m.global_bias == pytest.approx(mu)
    c = df.assign(c=df.rating - mu)
    bi = c.groupby("item_id").c.sum() / (c.groupby("item_id").c.count() + 5.0)
    assert m.item_biases == pytest.approx(bi.values.astype(np.float32))
    c["c2"] = c.c - c.item_id.map(bi)
    bu = c.groupby("user_id").c2.sum() / (c.groupby("user_id").c2.count() + 5.0)
    assert m.user_biases == pytest.approx(bu.values.astype(np.float32), rel=1e-5)
    rmat = ds.interaction_matrix(format="scipy", layout="coo", field="rating")
    t = m.transform_matrix(sps.coo_array(rmat))
    want = rmat.data - mu - m.item_biases[rmat.col] - m.user_biases[rmat.row]
    assert np.allclose(t.data, want)
    # scoring side: biases of items for a known user == what was subtracted
    b, ub = m.compute_for_items(ItemList([10, 20, 30]), 1)
    assert ub == pytest.approx(m.user_biases[0])
    assert b == pytest.approx(mu + m.item_biases + m.user_biases[0])
    # a user given by ratings: damped mean of the item-centred ratings
    b2, ub2 = m.compute_for_items(ItemList([10]), None,
                                  ItemList([10, 20], rating=np.array([5.0, 5.0])))
    exp = ((5 - mu - m.item_biases[0]) + (5 - mu - m.item_biases[1])) / (2 + 5.0)
    assert ub2 == pytest.approx(exp)
    assert m.compute_for_items(ItemList([10]), bias=0.25)[0] == pytest.approx(
        mu + m.item_biases[0] + 0.25)


def test_biased_mf_host_side(oracle):
    """The host half of the biased-MF component (src/lenskit/als/_explicit.py): config
    defaults, the unit-row init recipe, and ``finalize_scores`` adding b_g + b_i + b_u back."""
    from types import SimpleNamespace

    from lkpy_amd.als import BiasedMFConfig, BiasedMFScorer, BiasedMFTrainer
    from lkpy_amd.basic import BiasModel
    from lkpy_amd.data import ItemList, from_interactions_df

    cfg = BiasedMFConfig(features=20, regularization={"user": 0.2, "item": 0.05})
    assert cfg.embedding_size == 20 and cfg.user_reg == 0.2 and cfg.item_reg == 0.05
    assert cfg.damping == 5.0 and cfg.epochs == 10

    # initial_params: same stream, same arithmetic as the oracle's restatement
    fake = SimpleNamespace(rng=np.random.default_rng(7))
    got = BiasedMFTrainer.initial_params(fake, 50, 8)
    want = oracle.als_explicit_initial_params(np.random.default_rng(7), 50, 8)
    assert np.array_equal(got, want) and got.dtype == np.float32

    df = pd.DataFrame({"user_id": [1, 1, 2, 2, 3], "item_id": [10, 20, 10, 30, 20],
                       "rating": [4.0, 3.0, 2.0, 5.0, 1.0]})
    ds = from_interactions_df(df)
    sc = BiasedMFScorer()
    sc.bias = BiasModel.learn(ds, damping=5.0)
    sc.items, sc.users = ds.items, ds.users
    raw = ItemList([10, 20, 30], scores=np.array([0.5, -0.25, 0.0], np.float32))
    out = sc.finalize_scores(0, raw, None)  # stored user 0 -> its stored bias
    b = sc.bias
    want = raw.scores() + b.global_bias + b.item_biases + b.user_biases[0]
    assert np.allclose(out.scores(), want)
    out2 = sc.finalize_scores(None, raw, 0.75)  # fold-in supplies the user bias
    assert np.allclose(out2.scores(), raw.scores() + b.global_bias + b.item_biases + 0.75)
    out3 = sc.finalize_scores(None, raw, None)  # unknown user, no ratings: bias 0
    assert np.allclose(out3.scores(), raw.scores() + b.global_bias + b.item_biases)


def test_implicit_history_rows_drop_unknown_items():
    """Fold-in input of the batched kernel: histories become a CSR of
    confidence values, sorted by item number; items the model does not know are dropped, an
    empty or missing history gives an empty row."""
    from lkpy_amd.als import ImplicitMFConfig, ImplicitMFScorer
    from lkpy_amd.data import ItemList, RecQuery, Vocabulary

    sc = ImplicitMFScorer(ImplicitMFConfig(weight=40.0))
    sc.items = Vocabulary(np.array([10, 20, 30, 40]))
    qs = [RecQuery(user_items=ItemList([30, 999, 10])), RecQuery(user_id=5),
          RecQuery(user_items=ItemList([40]))]
    ptr, idx, val = sc._history_rows(qs)
    assert ptr.tolist() == [0, 2, 2, 3] and ptr.dtype == np.int64
    assert idx.tolist() == [0, 2, 3] and idx.dtype == np.int32  # sorted, 999 dropped
    assert val.tolist() == [40.0, 40.0, 40.0] and val.dtype == np.float32
    sc2 = ImplicitMFScorer(ImplicitMFConfig(weight=2.0, use_ratings=True))
    sc2.items = sc.items
    _, idx2, val2 = sc2._history_rows(
        [RecQuery(user_items=ItemList([20, 10], rating=np.array([3.0, 5.0])))])
    assert idx2.tolist() == [0, 1] and val2.tolist() == [10.0, 6.0]  # rating * weight, reordered
    with pytest.raises(ValueError):
        sc2._history_rows([RecQuery(user_items=ItemList([20]))])  # use_ratings without ratings


@pytest.mark.parametrize("scale", [0.04, 1.0])
def test_ml25m_like_meets_the_dataset_statistics(scale):
    """SURVEY.md 8d, cfg2-4 input: at full scale the public ML-25M counts EXACTLY -- nnz
    25 000 095, 162 541 x 62 423, 3 376 unrated items, rows of 20 .. 32 202, busiest item
    81 491; distinct sorted rows; deterministic in the seed."""
    from lkpy_amd import synth

    m = synth.ml25m_like(scale=scale)
    d = synth.describe(m)
    c = synth.ML25M
    assert d["nnz"] == int(c["nnz"] * scale)
    assert d["empty_items"] == int(c["n_empty_items"] * scale)
    if scale == 1.0:
        assert (d["n_users"], d["n_items"]) == (c["n_users"], c["n_items"])
        assert d["user_len_min"] == c["min_user"] and d["user_len_max"] == c["max_user"]
        assert d["item_len_max"] == c["max_item"]
    rows = np.repeat(np.arange(m.shape[0]), np.diff(m.indptr))
    same = rows[1:] == rows[:-1]
    assert np.all(np.diff(m.indices)[same] > 0)
    assert set(np.unique(m.data)) <= set(np.arange(1, 11, dtype=np.float32) * 0.5)
    if scale < 1.0:
        again = synth.ml25m_like(scale=scale)
        assert np.array_equal(again.indices, m.indices) and np.array_equal(again.data, m.data)


def test_user_knn_component_constructs_and_validates():
    "config mirror of src/lenskit/knn/user.py:39-71 (no GPU needed to build the component)"
    from lkpy_amd.knn import UserKNNConfig, UserKNNScorer

    u = UserKNNScorer(k=30, min_sim=1.0e-6)
    assert u.config.max_nbrs == 30 and u.config.explicit and not u.is_trained()
    assert UserKNNConfig(nnbrs=5, feedback="implicit").max_nbrs == 5
    assert UserKNNConfig(min_sim=1e-320).min_sim >= float(np.finfo(np.float64).smallest_normal)
    with pytest.raises(Exception):
        UserKNNConfig(bogus=1)


def test_legacy_pickle_state_is_migrated():
    """Scorers pickled before the learned arrays became lazily-synchronised descriptors keep them
    under their plain names; ``__setstate__`` moves them to the descriptor slots."""
    import pickle

    from lkpy_amd.als import BiasedMFScorer, ImplicitMFScorer
    from lkpy_amd.data import Vocabulary

    sc = ImplicitMFScorer(features=4)
    legacy = dict(sc.__getstate__())
    for k_ in ("_h_user_embeddings", "_h_item_embeddings", "_h__OtOr"):
        legacy.pop(k_, None)
    legacy.update(user_embeddings=np.ones((3, 4), np.float32),
                  item_embeddings=np.full((5, 4), 2.0, np.float32),
                  _OtOr=np.eye(4, dtype=np.float32), items=Vocabulary(np.arange(5), "item"))
    new = ImplicitMFScorer.__new__(ImplicitMFScorer)
    new.__setstate__(legacy)
    assert new.user_embeddings.shape == (3, 4) and new.item_embeddings[0, 0] == 2.0
    assert np.array_equal(new._OtOr, np.eye(4, dtype=np.float32))
    # and the current format still round-trips
    again = pickle.loads(pickle.dumps(new))
    assert np.array_equal(again.item_embeddings, new.item_embeddings)
    b = BiasedMFScorer.__new__(BiasedMFScorer)
    b.__setstate__({"config": BiasedMFScorer(features=4).config,
                    "user_embeddings": None, "item_embeddings": np.zeros((2, 4), np.float32)})
    assert b.user_embeddings is None and b.item_embeddings.shape == (2, 4)


def test_dataset_reports_repeated_pairs():
    "ADVICE r2: repeated (user, item) pairs must reach the kernels as ONE summed entry"
    from lkpy_amd.data import Dataset

    ds = Dataset.from_arrays([1, 1, 2, 1], [10, 11, 10, 10], [1.0, 2.0, 3.0, 4.0])
    assert ds.has_duplicates
    assert not Dataset.from_arrays([1, 1, 2], [10, 11, 10], [1.0, 2.0, 3.0]).has_duplicates
    m = ds.interactions().matrix().scipy(attribute="rating", layout="csr")
    m = m.copy()
    m.sum_duplicates()
    assert m.nnz == 3 and m[0, 0] == 5.0


def test_prepare_matrix_leaves_dataset_alone():
    """``prepare_matrix`` canonicalised repeated pairs IN the Dataset's own
    index arrays (``scipy()`` hands them out without a copy; ``sum_duplicates`` rewrites them in
    place).  The dataset must be untouched and the matrix must hold the summed entry."""
    from lkpy_amd.als import ImplicitMFScorer, ImplicitMFTrainer
    from lkpy_amd.data import Dataset

    ds = Dataset.from_arrays([1, 1, 1, 1, 2, 2, 3], [10, 11, 11, 12, 10, 10, 12],
                             [1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0])
    assert ds.has_duplicates
    cols, ptr, rows = ds._cols.copy(), ds._indptr.copy(), ds._rows.copy()
    attrs = {k: v.copy() for k, v in ds._attrs.items()}
    tr = ImplicitMFTrainer.__new__(ImplicitMFTrainer)
    tr.scorer = ImplicitMFScorer(embedding_size=4, weight=2.0, use_ratings=True)
    m = tr.prepare_matrix(ds)
    assert m.nnz == 5 and m.has_canonical_format
    assert m[0, 1] == 2.0 * (2.0 + 3.0) and m[1, 0] == 2.0 * (5.0 + 6.0)
    assert np.array_equal(ds._cols, cols) and np.array_equal(ds._indptr, ptr)
    assert np.array_equal(ds._rows, rows)
    for k, v in attrs.items():
        assert np.array_equal(ds._attrs[k], v)
    # a second model trained on the same dataset sees the same matrix
    m2 = tr.prepare_matrix(ds)
    assert (m != m2).nnz == 0
    # without repeated pairs nothing is copied or summed
    ds2 = Dataset.from_arrays([1, 1, 2], [10, 11, 10], [1.0, 2.0, 3.0])
    tr.scorer = ImplicitMFScorer(embedding_size=4)
    m3 = tr.prepare_matrix(ds2)
    assert m3.nnz == 3 and np.all(m3.data == 40.0)


def test_load_movielens_zip_and_directory(tmp_path):
    """``load_movielens`` (src/lenskit/data/sources/movielens.py:327-345,435-452): a zip named like
    the reference's fixture path (``data/ml-25m.zip``, testing/_movielens.py:34) and an unpacked
    directory; items = every movies.csv id (unrated movies are empty rows) plus ids only the
    ratings mention; interactions (user, item)-sorted; float32 ratings."""
    import zipfile

    from lkpy_amd.data import load_movielens, load_movielens_df

    movies = "movieId,title,genres\n1,A (1990),Drama\n2,B,Comedy\n5,\"C, the\",(no genres listed)\n9,D,Drama\n"
    ratings = ("userId,movieId,rating,timestamp\n7,5,4.5,100\n3,1,3.0,101\n7,1,0.5,102\n"
               "3,9,5.0,103\n11,12,2.0,104\n")
    z = tmp_path / "ml-25m.zip"
    with zipfile.ZipFile(z, "w") as zf:
        zf.writestr("ml-25m/", "")
        zf.writestr("ml-25m/movies.csv", movies)
        zf.writestr("ml-25m/ratings.csv", ratings)
    d = tmp_path / "ml-latest-small"
    d.mkdir()
    (d / "movies.csv").write_text(movies)
    (d / "ratings.csv").write_text(ratings)
    for src in (z, d):
        ds = load_movielens(src)
        assert list(ds.users._ids) == [3, 7, 11]
        assert list(ds.items._ids) == [1, 2, 5, 9, 12]  # movie 2 unrated, 12 inserted
        m = ds.interactions().matrix().scipy(attribute="rating", layout="csr")
        assert m.dtype == np.float32 and m.shape == (3, 5)
        assert m.toarray().tolist() == [[3.0, 0, 0, 5.0, 0], [0.5, 0, 4.5, 0, 0], [0, 0, 0, 0, 2.0]]
        assert not ds.has_duplicates
        df = load_movielens_df(src)
        assert list(df.columns) == ["user_id", "item_id", "rating", "timestamp"]
        assert df["user_id"].dtype == np.int32 and df["rating"].dtype == np.float32
    # an unnamed directory is detected by its files (movielens.py:510-529)
    other = tmp_path / "somewhere"
    d.rename(other)
    assert load_movielens(other).interaction_count == 5
    with pytest.raises(RuntimeError):
        load_movielens(tmp_path / "absent")


def test_load_movielens_equals_the_committed_fixture():
    "the loader on the reference checkout's ml-latest-small == tests/golden/ml_small.npz"
    from pathlib import Path

    from lkpy_amd.data import load_movielens, load_movielens_npz

    src = Path("/root/reference/data/ml-latest-small")
    if not src.exists():
        pytest.skip("reference checkout not present (GPU box)")
    ds = load_movielens(src)
    g = load_movielens_npz(Path(__file__).parent / "golden" / "ml_small.npz")
    assert ds.users == g.users and ds.items == g.items
    assert np.array_equal(ds._cols, g._cols) and np.array_equal(ds._indptr, g._indptr)
    assert np.array_equal(ds._attrs["rating"], g._attrs["rating"])


def test_item_list_collection_mirrors_the_reference_api():
    """``batch.recommend`` / ``batch.predict`` hand back an ``ItemListCollection`` keyed by
    ``user_id`` (src/lenskit/data/_collection/_base.py:48-592, batch/_runner.py:157-191)."""
    from lkpy_amd.data import ItemList, ItemListCollection

    c = ItemListCollection.from_dict(
        {3: ItemList([1, 2, 3], scores=[0.3, 0.2, 0.1], ordered=True), 5: ItemList([7])})
    assert len(c) == 2 and c.key_fields == ("user_id",) and c.total_items() == 4
    assert c.lookup(3).ids().tolist() == [1, 2, 3]
    assert c.lookup(user_id=5) is c.lookup((5,)) and c.lookup(9) is None
    key, il = c[1]  # positional, like the reference
    assert key.user_id == 5 and len(il) == 1
    assert [k.user_id for k in c.keys()] == [3, 5] and [len(x) for x in c.lists()] == [3, 1]
    assert [(k.user_id, len(x)) for k, x in c] == [(3, 3), (5, 1)]
    df = c.to_df()
    assert list(df.columns[:2]) == ["user_id", "item_id"] and len(df) == 4
    assert df[df.user_id == 3]["rank"].tolist() == [1, 2, 3]
    c.add(ItemList([9]), 8)
    assert c.lookup(8).ids().tolist() == [9]
    # equal keys: both lists are kept, ``lookup`` gives the LAST one -- the reference's ListILC
    # (src/lenskit/data/_collection/_list.py:190-193, 203-227)
    c.add(ItemList([10, 11]), 8)
    assert len(c) == 4 and c.lookup(8).ids().tolist() == [10, 11]
    assert [k.user_id for k in c.keys()] == [3, 5, 8, 8]
    two = ItemListCollection(("user_id", "seq"))
    two.add(ItemList([1]), 4, 0)
    two.add(ItemList([2]), user_id=4, seq=1)
    assert two.lookup(4, 1).ids().tolist() == [2] and two.key_type._fields == ("user_id", "seq")


def test_array_backed_collection_builds_lists_on_demand():
    """``ItemListCollection.from_arrays``: the [B x n] arrays of a batched recommend call as a
    collection; keys, lists and the lookup index are only built when asked for."""
    from lkpy_amd.data import ItemList, ItemListCollection, Vocabulary

    v = Vocabulary(np.arange(100, 120), "item")
    nums = np.array([[3, 2, -1], [5, 6, 7], [-1, -1, -1]], np.int32)
    sc = np.array([[.9, .8, np.nan], [.5, .4, .3], [np.nan] * 3], np.float32)
    c = ItemListCollection.from_arrays(np.array([11, 12, 13]), nums, sc, v)
    assert len(c) == 3 and c.total_items() == 5 and not c._lists._made
    assert [k.user_id for k in c.keys()] == [11, 12, 13] and isinstance(c[0][0].user_id, int)
    assert c.lookup(11).ids().tolist() == [103, 102] and c.lookup(11).ordered
    assert c.lookup(user_id=12).scores().tolist() == sc[1].tolist()
    assert len(c.lookup(13)) == 0 and c.lookup(99) is None
    assert c[1][0].user_id == 12 and c[-1][0].user_id == 13
    df = c.to_df()
    assert df["user_id"].tolist() == [11, 11, 12, 12, 12]
    assert df["item_id"].tolist() == [103, 102, 105, 106, 107] and df["rank"].tolist() == [1, 2, 1, 2, 3]
    c.add(ItemList(item_nums=[1], vocabulary=v, scores=[1.0]), 12)  # the last of equal keys wins
    assert len(c) == 4 and c.lookup(12).ids().tolist() == [101]
    assert [k.user_id for k, _ in c] == [11, 12, 13, 12] and c.total_items() == 6


def test_history_batch_host_side():
    """``UserTrainingHistoryLookup.batch``: user numbers by one vocabulary lookup, unknown users
    as empty histories (src/lenskit/basic/history.py:77-95), string ids for a numeric vocabulary
    converted (85-87), ``queries()`` == the per-query lookup."""
    from lkpy_amd.basic import UserTrainingHistoryLookup
    from lkpy_amd.data import Dataset

    ds = Dataset.from_arrays([10, 10, 20, 30, 30, 30], [1, 2, 2, 1, 2, 3],
                             [4.0, 3.0, 5.0, 1.0, 2.0, 3.0])
    lk = UserTrainingHistoryLookup()
    lk.train(ds)
    hb = lk.batch([30, 99, 10, "20"])
    assert len(hb) == 4 and hb.user_nums.tolist() == [2, -1, 0, 1]
    assert hb.lengths.tolist() == [3, 0, 2, 1] and hb.has_ratings and hb.items is ds.items
    qs = hb.queries()
    assert [q.user_id for q in qs] == [30, 99, 10, 20]
    assert qs[0].query_items.ids().tolist() == [1, 2, 3] and qs[1].query_items is None
    sub = hb.subset(np.array([True, False, True, False]))
    assert sub.user_nums.tolist() == [2, 0] and sub.lengths.tolist() == [3, 2]
    import pickle

    assert pickle.loads(pickle.dumps(lk)).batch([10]).lengths.tolist() == [2]
