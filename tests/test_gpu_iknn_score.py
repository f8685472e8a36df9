"""GPU parity: batched item-kNN scoring vs the oracle (one query at a time on the CPU)."""
import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sps
import torch
from pathlib import Path

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).parent / "golden"


def _dev_lists(lists, dtype, gpu):
    ptr = np.zeros(len(lists) + 1, np.int64)
    np.cumsum([len(x) for x in lists], out=ptr[1:])
    flat = np.concatenate(lists).astype(dtype) if len(lists) else np.zeros(0, dtype)
    return torch.from_numpy(ptr).to(gpu), torch.from_numpy(flat).to(gpu), ptr


@pytest.fixture(scope="module")
def model(oracle, ml_small, gpu):
    from lkpy_amd import _device as D

    ui, iu, means, _ = oracle.iknn_prepare(ml_small["rmat"], True)
    sims = oracle.iknn_build(ui, iu, 1.0e-6, None)
    dsims = D.DeviceCSR(torch.from_numpy(sims.indptr.astype(np.int64)).to(gpu),
                        torch.from_numpy(sims.indices.astype(np.int32)).to(gpu),
                        torch.from_numpy(sims.data.astype(np.float32)).to(gpu), sims.shape, None)
    return sims, dsims, means


@pytest.mark.parametrize("lists", [True, False])
@pytest.mark.parametrize("explicit", [True, False])
@pytest.mark.parametrize("max_nbrs,min_nbrs", [(20, 1), (5, 3), (100, 1)])
def test_score_batch_matches_oracle(gpu, oracle, ml_small, model, rng, monkeypatch, explicit,
                                    max_nbrs, min_nbrs, lists):
    """Both kernels -- the candidate-list kernel and, with LK_KNN_SCORE_LISTS=0, the slot kernel --
    follow the reference's accumulator step for step (the vector, then std's BinaryHeap push /
    pop: the same one of several EQUAL similarities is evicted at the max_nbrs boundary, the sums
    run over the same array order): the scores are the oracle's, bit for bit."""
    from lkpy_amd import _device as D

    monkeypatch.setenv("LK_KNN_SCORE_LISTS", "1" if lists else "0")
    sims, dsims, means = model
    csr = sps.csr_array(ml_small["rmat"])
    users = rng.choice(csr.shape[0], 40, replace=False)
    hists, rates, tgts = [], [], []
    for u in users:
        h = csr.indices[csr.indptr[u] : csr.indptr[u + 1]].astype(np.int32)
        r = csr.data[csr.indptr[u] : csr.indptr[u + 1]].astype(np.float32) - means[h]
        if u % 5 == 0:  # unknown history items (nulls) are skipped
            h = np.concatenate([h, [-1, -1]]).astype(np.int32)
            r = np.concatenate([r, [3.0, -2.0]]).astype(np.float32)
        t = rng.choice(csr.shape[1], 300, replace=False).astype(np.int32)
        t[::50] = -1  # null targets
        hists.append(h), rates.append(r), tgts.append(t)
    hists[3] = np.zeros(0, np.int32)  # a query without history
    rates[3] = np.zeros(0, np.float32)
    rp, ri, _ = _dev_lists(hists, np.int32, gpu)
    _, rr, _ = _dev_lists(rates, np.float32, gpu)
    tp, ti, tptr = _dev_lists(tgts, np.int32, gpu)
    gs, gc = D.iknn_score_batch(dsims, rp, ri, rr if explicit else None, tp, ti, max_nbrs,
                                min_nbrs)
    gs, gc = gs.cpu().numpy(), gc.cpu().numpy()
    n_list, n_slot, nt_max = D.knn_score_last_stats()
    assert (n_list, n_slot) == ((len(users), 0) if lists else (0, len(users)))
    assert nt_max == 300 or not lists
    n_bits = n_ok = 0
    for q in range(len(users)):
        ws_, wc = oracle.iknn_score(sims, hists[q], rates[q] if explicit else None, tgts[q],
                                    max_nbrs, min_nbrs)
        s, c = gs[tptr[q] : tptr[q + 1]], gc[tptr[q] : tptr[q + 1]]
        assert np.array_equal(c, wc)  # neighbour counts are integer work: exact
        assert np.array_equal(np.isnan(s), np.isnan(ws_))
        ok = ~np.isnan(ws_)
        err = np.abs(s - ws_) / np.maximum(np.abs(ws_), 1e-3)
        n_ok += int(ok.sum())
        n_bits += int(np.sum(s[ok].view(np.uint32) == ws_[ok].view(np.uint32)))
        assert np.all(err[ok] <= 1e-5), (q, float(err[ok].max()))
    print(f"\n{'list' if lists else 'slot'} kernel: scores bit-identical to the oracle's "
          f"{n_bits} of {n_ok}")
    assert n_bits == n_ok  # same accumulator steps, same summation order, same roundings


def test_list_kernel_repeats_long_lists_and_rounds(gpu, oracle, ml_small, model, rng, monkeypatch):
    """The candidate-list kernel's corners: a target item named more than once, exactly 1024
    targets (the limit; 1025 go to the slot kernel), the heaviest histories (several rounds of
    256 history rows, the accumulators carried between them) on the most popular targets (lists
    far past max_nbrs: long runs of heap pushes and pops), max_nbrs = 255 (the limit) and 256."""
    from lkpy_amd import _device as D

    sims, dsims, means = model
    csr = sps.csr_array(ml_small["rmat"])
    pop = np.argsort(-np.diff(sps.csc_array(csr).indptr), kind="stable")
    heavy = np.argsort(-np.diff(csr.indptr), kind="stable")[:3]
    light = np.argsort(np.diff(csr.indptr), kind="stable")[:5]

    def hist(u):
        h = csr.indices[csr.indptr[u] : csr.indptr[u + 1]].astype(np.int32)
        return h, csr.data[csr.indptr[u] : csr.indptr[u + 1]].astype(np.float32) - means[h]

    def run(users, tgts, max_nbrs, min_nbrs):
        hs, rs = zip(*[hist(u) for u in users])
        rp, ri, _ = _dev_lists(list(hs), np.int32, gpu)
        _, rr, _ = _dev_lists(list(rs), np.float32, gpu)
        tp, ti, tptr = _dev_lists(tgts, np.int32, gpu)
        gs, gc = D.iknn_score_batch(dsims, rp, ri, rr, tp, ti, max_nbrs, min_nbrs)
        gs, gc = gs.cpu().numpy(), gc.cpu().numpy()
        stats = D.knn_score_last_stats()
        for q in range(len(users)):
            ws_, wc = oracle.iknn_score(sims, hs[q], rs[q], tgts[q], max_nbrs, min_nbrs)
            s, c = gs[tptr[q] : tptr[q + 1]], gc[tptr[q] : tptr[q + 1]]
            assert np.array_equal(c, wc)
            assert np.array_equal(np.isnan(s), np.isnan(ws_))
            ok = ~np.isnan(ws_)
            assert np.array_equal(s[ok].view(np.uint32), ws_[ok].view(np.uint32))
        return stats

    # repeats + nulls
    t = rng.choice(csr.shape[1], 200, replace=False).astype(np.int32)
    t = np.concatenate([t, t[:50], [-1], t[10:20]]).astype(np.int32)
    assert run(light, [t] * len(light), 20, 2)[:2] == (len(light), 0)
    # 1024 targets: still the list kernel; 1025: slot kernel
    t1024 = pop[:1024].astype(np.int32)
    st = run(light[:2], [t1024, t1024[:7]], 20, 1)
    assert st == (2, 0, 1024)
    assert run(light[:2], [pop[:1025].astype(np.int32), t1024[:7]], 20, 1)[:2] == (0, 2)
    # the heaviest histories (2 698, 1 864, 1 291 items: 11 / 8 / 6 rounds) x popular targets;
    # with LK_KNN_SCORE_HEAVY=1000 all three are split into parts by target as well
    tp_ = pop[:100].astype(np.int32)
    for mn in (20, 100, 255):
        assert run(list(heavy) + list(light[:2]), [tp_] * 5, mn, 1)[:2] == (5, 0)
    monkeypatch.setenv("LK_KNN_SCORE_HEAVY", "1000")
    for split in ("2", "8", "64"):
        monkeypatch.setenv("LK_KNN_SCORE_SPLIT", split)
        assert run(list(heavy) + list(light[:2]), [tp_, t, tp_, t, tp_], 100, 1)[:2] == (5, 0)
    monkeypatch.delenv("LK_KNN_SCORE_HEAVY")
    monkeypatch.delenv("LK_KNN_SCORE_SPLIT")
    assert run(list(heavy[:1]), [tp_], 256, 1)[:2] == (0, 1)  # past the list kernel's limit
    assert len(hist(heavy[0])[0]) > 4 * 256


def test_known_preds_golden(gpu, oracle, ml_small, model):
    """tests/models/test_knn_item_item.py:413-453 through the GPU scorer."""
    from lkpy_amd import _device as D

    sims, dsims, means = model
    known = pd.read_csv(GOLDEN / "item-item-preds.csv")
    csr = sps.csr_array(ml_small["rmat"])
    hists, rates, tgts, exp = [], [], [], []
    for uid, grp in known.groupby("user_id"):
        u = int(np.searchsorted(ml_small["user_ids"], uid))
        h = csr.indices[csr.indptr[u] : csr.indptr[u + 1]].astype(np.int32)
        hists.append(h)
        rates.append(csr.data[csr.indptr[u] : csr.indptr[u + 1]].astype(np.float32) - means[h])
        tgts.append(np.searchsorted(ml_small["item_ids"], grp.item_id.values).astype(np.int32))
        exp.append(grp.prediction.values)
    rp, ri, _ = _dev_lists(hists, np.int32, gpu)
    _, rr, _ = _dev_lists(rates, np.float32, gpu)
    tp, ti, _ = _dev_lists(tgts, np.int32, gpu)
    gs, gc = D.iknn_score_batch(dsims, rp, ri, rr, tp, ti, 20, 1)
    pred = gs.cpu().numpy() + means[np.concatenate(tgts)]
    exp = np.concatenate(exp)
    assert not np.any(np.isnan(pred) & ~np.isnan(exp))  # the reference's hard assertion (line 435)
    err = np.abs(pred - exp)
    assert np.sum(err > 1e-5) <= 5 and np.median(err) < 1e-6


def test_nan_similarity_is_an_error(gpu):
    from lkpy_amd import _device as D

    ptr = torch.tensor([0, 1, 2], dtype=torch.int64, device=gpu)
    idx = torch.tensor([1, 0], dtype=torch.int32, device=gpu)
    val = torch.tensor([float("nan"), 0.5], dtype=torch.float32, device=gpu)
    sims = D.DeviceCSR(ptr, idx, val, (2, 2), None)
    one = torch.tensor([0, 1], dtype=torch.int64, device=gpu)
    with pytest.raises(ValueError, match="similarity is null"):
        D.iknn_score_batch(sims, one, torch.tensor([0], dtype=torch.int32, device=gpu), None, one,
                           torch.tensor([1], dtype=torch.int32, device=gpu), 5, 1)
