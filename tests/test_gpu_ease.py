"""
EASE on the GPU against the oracle's restatement of
src/lenskit/knn/ease.py: Gramian, trained weights, batch scores, component behaviour.
"""
import pickle

import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu


def _rand_binary(rng, n_users, n_items, density):
    m = sps.random(n_users, n_items, density=density, random_state=rng, format="csr",
                   dtype=np.float32)
    m.data[:] = 1.0
    return sps.csr_array(m)


def test_ease_gram_exact(gpu, rng):
    import torch

    from lkpy_amd import _device as D

    ui = _rand_binary(rng, 300, 130, 0.08)
    iu = sps.csr_array(ui.T)
    iu.sort_indices()
    cooc = D.iknn_build(D.DeviceCSR.from_scipy(ui, gpu), D.DeviceCSR.from_scipy(iu, gpu), 0.5)
    counts = torch.from_numpy(np.diff(iu.indptr).astype(np.int32)).to(gpu)
    g = D.ease_gram(cooc, counts, 2.5).cpu().numpy()
    want = np.asarray((ui.T @ ui).todense(), dtype=np.float32)
    want[np.diag_indices(130)] += np.float32(2.5)
    assert np.array_equal(g, want)  # integer counts: exact


@pytest.mark.parametrize("reg", [1.0, 50.0])
def test_ease_weights_and_scores_vs_oracle(gpu, oracle, rng, reg):
    from lkpy_amd.data import Dataset, ItemList, RecQuery
    from lkpy_amd.knn import EASEScorer

    ui = _rand_binary(rng, 400, 150, 0.06)
    coo = ui.tocoo()
    ds = Dataset.from_arrays(coo.row + 1000, coo.col + 10, np.ones(coo.nnz, np.float32))
    algo = EASEScorer(regularization=reg)
    algo.train(ds)
    assert algo.is_trained()
    # the oracle on the matrix in the dataset's own numbering
    mat = ds.interactions().matrix().scipy(attribute=None).astype(np.float32)
    want = oracle.ease_train(sps.csr_array(mat), reg)
    assert algo.weights.shape == want.shape
    assert np.all(np.diag(algo.weights) == 0.0)
    rel = np.linalg.norm(algo.weights - want) / np.linalg.norm(want)
    assert rel < 1e-4, rel

    # batch scoring = the reference's q_vec @ weights per query
    items = ItemList(item_ids=ds.items.ids())
    hist_nums = [np.array([3, 7, 11], np.int32), np.array([0], np.int32),
                 np.arange(0, 60, 2, dtype=np.int32)]
    queries = [RecQuery(user_items=ItemList(item_ids=ds.items.ids(h)))
               for h in hist_nums]
    outs = algo.score_batch(queries, [items] * len(queries))
    for h, out in zip(hist_nums, outs):
        ref = oracle.ease_score(algo.weights, h)
        got = out.scores()
        assert np.allclose(got, ref, rtol=1e-5, atol=1e-6)
    # single query == batch
    one = algo(queries[0], items).scores()
    assert np.array_equal(one, outs[0].scores())
    # no history / unknown items: all NaN, unknown targets NaN (ease.py:150-158, 170)
    nohist = algo(RecQuery(user_id=5), items)
    assert np.all(np.isnan(nohist.scores()))
    mixed = ItemList(item_ids=np.concatenate([ds.items.ids()[:3], [987654]]))
    sc = algo(queries[0], mixed).scores()
    assert np.isnan(sc[3]) and not np.isnan(sc[:3]).any()

    # pickle round trip (device state is rebuilt lazily)
    clone = pickle.loads(pickle.dumps(algo))
    assert np.array_equal(clone(queries[2], items).scores(), outs[2].scores())


def test_ease_config_errors(gpu):
    from lkpy_amd.data import Dataset
    from lkpy_amd.knn import EASEScorer
    from lkpy_amd.training import TrainingOptions

    with pytest.raises(Exception):
        EASEScorer(regularization=0.0)
    ds = Dataset.from_arrays(np.array([1, 1, 2]), np.array([5, 6, 5]), np.ones(3, np.float32))
    with pytest.raises(ValueError):
        EASEScorer().train(ds, TrainingOptions(environment={"LK_EASE_SOLVER": "magic"}))
