"""GPU parity: dense scoring + top-N vs the oracle -- scores and index lists BIT-EXACT."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k", [8, 25, 64, 128])
def test_score_topk_bit_exact(gpu, oracle, rng, k):
    from lkpy_amd import _device as D

    B, I, n = 150, 3000, 100
    U = rng.standard_normal((B, k)).astype(np.float32)
    Q = rng.standard_normal((I, k)).astype(np.float32)
    Q[rng.random(I) < 0.05] = 0.0  # unrated items: zero factors (score 0)
    # exclusion lists = the query's own items (basic/candidates.py:77-94)
    lens = rng.integers(0, 80, B)
    lens[0] = 0
    ptr = np.zeros(B + 1, np.int64)
    np.cumsum(lens, out=ptr[1:])
    ex = np.concatenate([np.sort(rng.choice(I, l, replace=False)) for l in lens]).astype(np.int32)
    idx, sc = D.score_topk(
        D.to_device_padded(U, gpu), D.to_device_padded(Q, gpu), k, n,
        torch.from_numpy(ptr).to(gpu), torch.from_numpy(ex).to(gpu))  # fmt: skip
    idx, sc = idx.cpu().numpy(), sc.cpu().numpy()
    for b in range(B):
        s = oracle.score_dense(Q, U[b])  # fixed k-ordered fmaf chain
        s[ex[ptr[b] : ptr[b + 1]]] = np.nan
        want = oracle.argtopn(s, n)  # the reference's heap (sorting.rs:132-172)
        assert np.array_equal(sc[b].view(np.uint32), s[want].view(np.uint32)), b
        # continuous random scores: no ties, so the index lists must agree exactly
        assert np.array_equal(idx[b], want), b
        assert not np.isin(idx[b], ex[ptr[b] : ptr[b + 1]]).any()


def test_argtopn_properties_and_ties(gpu, oracle, rng):
    """tests/accel/test_argsort.py:60-211 on the kernel, incl. NaN, short rows, heavy ties."""
    from lkpy_amd import _device as D

    rows, ln = 64, 5000
    s = np.round(rng.standard_normal((rows, ln)), 1).astype(np.float32)  # many ties
    s[rng.random((rows, ln)) < 0.1] = np.nan
    s[0, :] = np.nan  # nothing valid
    s[1, 50:] = np.nan  # fewer valid entries than n
    s[2, :] = 1.0  # all tied -> the lowest indices
    s[3, ::2] = -0.0
    s[3, 1::2] = 0.0
    for n in (1, 10, 100, 777):
        got = D.argtopn(torch.from_numpy(s).to(gpu), n).cpu().numpy()
        for r in range(rows):
            valid = ~np.isnan(s[r])
            g = got[r]
            m = min(n, int(valid.sum()))
            assert np.all(g[m:] == -1) and np.all(g[:m] >= 0)
            g = g[:m]
            assert len(set(g.tolist())) == m
            assert np.all(np.diff(s[r][g]) <= 0)  # descending
            if m:
                rest = np.setdiff1d(np.flatnonzero(valid), g)
                assert not np.any(s[r][rest] > s[r][g].min())  # nothing excluded beats the min
            # same multiset of scores as the reference heap, ties broken by lower index
            want = oracle.argtopn(s[r], n)
            assert np.array_equal(np.sort(s[r][want]), np.sort(s[r][g]))
            expect = np.lexsort((np.arange(ln)[valid], -(s[r][valid] + 0.0)))[:m]
            assert np.array_equal(g, np.flatnonzero(valid)[expect])
    assert np.array_equal(D.argtopn(torch.from_numpy(s).to(gpu), 5).cpu().numpy()[2], np.arange(5))


def test_argtopn_unaligned_rows(gpu, oracle, rng):
    "Odd row length: most rows start off a 16-byte boundary (scalar-load path of the kernel)."
    from lkpy_amd import _device as D

    rows, ln = 33, 4099
    s = rng.standard_normal((rows, ln)).astype(np.float32)
    s[rng.random((rows, ln)) < 0.05] = np.nan
    for n in (7, 100, 256, 257):
        got = D.argtopn(torch.from_numpy(s).to(gpu), n).cpu().numpy()
        for r in range(rows):
            want = oracle.argtopn(s[r], n)
            # same scores position by position; an exact tie across the cut may be resolved
            # differently (kernel: lower index; the reference heap: unspecified)
            assert np.array_equal(s[r][got[r]].view(np.uint32), s[r][want].view(np.uint32)), (n, r)
            assert len(set(got[r].tolist())) == n
            differ = got[r] != want
            assert np.all(np.isin(s[r][got[r]][differ], s[r][want][differ]))


def test_score_topk_cfg1_recommendations(gpu, oracle, ml_small):
    """cfg1 end of pipe: top-10 for every ml-latest-small user from oracle-trained factors,
    history excluded -- identical lists to scoring + the reference heap on the CPU."""
    import scipy.sparse as sps

    from lkpy_amd import _device as D

    rmat = ml_small["rmat"]
    ind = sps.coo_array((np.ones(rmat.nnz, np.float32), (rmat.row, rmat.col)), rmat.shape)
    st = oracle.als_train(ind, 25, 3, 7)
    P, Q = st.user_embeddings, st.item_embeddings
    csr = sps.csr_array(ind)
    csr.sort_indices()
    idx, sc = D.score_topk(
        D.to_device_padded(P, gpu), D.to_device_padded(Q, gpu), 25, 10,
        torch.from_numpy(csr.indptr.astype(np.int64)).to(gpu),
        torch.from_numpy(csr.indices.astype(np.int32)).to(gpu))  # fmt: skip
    idx = idx.cpu().numpy()
    for u in range(0, P.shape[0], 7):
        s = oracle.score_dense(Q, P[u])
        s[csr.indices[csr.indptr[u] : csr.indptr[u + 1]]] = np.nan
        want = oracle.argtopn(s, 10)
        if len(np.unique(s[want])) == 10 and s[want][-1] > np.sort(s[~np.isnan(s)])[-11]:
            assert np.array_equal(idx[u], want), u
        assert np.array_equal(np.sort(s[idx[u]]), np.sort(s[want]))


def _oracle_topn(oracle, Q, u, excl, n):
    s = oracle.score_dense(Q, u)
    s[excl] = np.nan
    want = oracle.argsort_descending(s)[:n]  # (score desc, index asc): the kernels' tie rule
    return s, want


ROUND4 = ("panel", "sort")  # LK_TOPK_STAGE1 / LK_TOPK_SELECT: the kernels round 5 replaced
ROUND5 = ("cmax", "wave")   # the defaults: class maxima from the sample GEMM, a wave per row


def _variant(monkeypatch, variant):
    monkeypatch.setenv("LK_TOPK_STAGE1", variant[0])
    monkeypatch.setenv("LK_TOPK_SELECT", variant[1])


@pytest.mark.parametrize("variant", [ROUND5, ROUND4, ("cmax", "sort")])
@pytest.mark.parametrize("k,n", [(64, 100), (25, 10), (128, 128), (10, 10)])
def test_score_topk_fused_path(gpu, oracle, rng, k, n, monkeypatch, variant):
    """
    The fused scoring + selection path (catalogues >= 16 384 items: stage-1 threshold from an
    item sample, stage-2 GEMM emitting candidates only, stage-3 exact order) gives the SAME
    lists as the panel path and as the oracle: exclusion lists in any order and of any length,
    exact ties across the threshold, rows with fewer than n candidates, rows that overflow the
    candidate buffer (all scores equal) and fall back.  (k = 10 pads to 16 features, less than a
    staged slab: such calls take the panel path whatever the sizes.)
    """
    from lkpy_amd import _device as D

    _variant(monkeypatch, variant)
    monkeypatch.setenv("LK_TOPK_FUSED_MIN_USERS", "1")
    B, I = 200, 20000
    U = rng.standard_normal((B, k)).astype(np.float32)
    Q = rng.standard_normal((I, k)).astype(np.float32)
    Q[rng.random(I) < 0.05] = 0.0
    Q[5000:5200] = Q[100:300]  # exact score ties between distant items, for every user
    U[3] = 0.0  # every score 0: all tied -> candidate overflow -> panel fallback for the batch
    lens = rng.integers(0, 300, B)
    lens[0] = 0
    lens[1] = I - 40  # fewer than n candidates remain
    lens[2] = 15000  # a long list
    ptr = np.zeros(B + 1, np.int64)
    np.cumsum(lens, out=ptr[1:])
    ex = np.concatenate([rng.choice(I, l, replace=False) for l in lens]).astype(np.int32)  # unsorted
    dU, dQ = D.to_device_padded(U, gpu), D.to_device_padded(Q, gpu)
    dptr, dex = torch.from_numpy(ptr).to(gpu), torch.from_numpy(ex).to(gpu)
    idx, sc = D.score_topk(dU, dQ, k, n, dptr, dex)
    idx, sc = idx.cpu().numpy(), sc.cpu().numpy()
    for b in list(range(12)) + list(rng.choice(B, 30, replace=False)):
        s, want = _oracle_topn(oracle, Q, U[b], ex[ptr[b] : ptr[b + 1]], n)
        m = len(want)
        assert np.array_equal(idx[b, :m], want), b
        assert np.all(idx[b, m:] == -1) and np.all(np.isnan(sc[b, m:]))
        assert np.array_equal(sc[b, :m].view(np.uint32), s[want].view(np.uint32)), b
    # and the panel path (fused path switched off) agrees on every row, bit for bit
    monkeypatch.setenv("LK_TOPK_FUSED_MIN_ITEMS", "1000000000")
    idx2, sc2 = D.score_topk(dU, dQ, k, n, dptr, dex)
    assert np.array_equal(idx2.cpu().numpy(), idx)
    assert np.array_equal(sc2.cpu().numpy().view(np.uint32), sc.view(np.uint32))


def test_score_topk_fused_no_overflow_rows(gpu, oracle, rng, monkeypatch):
    "a clean batch (no fallback): scores without exclusions, n = 100"
    from lkpy_amd import _device as D

    monkeypatch.setenv("LK_TOPK_FUSED_MIN_USERS", "1")
    B, I, k, n = 70, 33000, 64, 100
    U = rng.standard_normal((B, k)).astype(np.float32)
    Q = (rng.standard_normal((I, k)) * rng.random((I, 1))).astype(np.float32)
    idx, sc = D.score_topk(D.to_device_padded(U, gpu), D.to_device_padded(Q, gpu), k, n)
    idx, sc = idx.cpu().numpy(), sc.cpu().numpy()
    for b in range(0, B, 7):
        s, want = _oracle_topn(oracle, Q, U[b], np.zeros(0, np.int64), n)
        assert np.array_equal(idx[b], want) and np.array_equal(sc[b], s[want])


@pytest.mark.parametrize("variant", [ROUND5, ROUND4])
@pytest.mark.parametrize("exact", [False, True])
def test_score_topk_fused_threshold_misses_are_redone(gpu, oracle, rng, monkeypatch, exact, variant):
    """
    The fused path's threshold is a RANK of the strided item sample chosen so that "fewer than n
    items reach it" is a 1e-6 event -- under random sampling.  An adversarial catalogue (every
    high-scoring item sits at an id that is a multiple of the stride, i.e. IN the sample) makes
    the event certain: such rows must be detected and redone exactly.  LK_TOPK_TAU_EXACT=1
    (threshold = n-th best of the sample, a certain bound) gives the same lists.
    """
    from lkpy_amd import _device as D

    _variant(monkeypatch, variant)
    monkeypatch.setenv("LK_TOPK_FUSED_MIN_USERS", "1")
    div = 24  # LK_TOPK_SAMPLE_DIV_DEFAULT
    if exact:
        # r = n at a sample of 1/16: ~1600 candidates per row -- the second selection tier
        # (more than 1024, at most 2048 candidates) takes the rows
        monkeypatch.setenv("LK_TOPK_TAU_EXACT", "1")
        div = 16
        monkeypatch.setenv("LK_TOPK_SAMPLE_DIV", "16")
    B, I, k, n = 96, 40000, 32, 100
    U = np.abs(rng.standard_normal((B, k))).astype(np.float32)
    Q = (rng.standard_normal((I, k)) * 0.01).astype(np.float32)
    # the sample is every (I // sub)-th item with sub = max(I / div, 16 n) rounded up to 256
    sub = max(I // div, 16 * n)
    sub = (sub + 255) // 256 * 256
    stride = I // sub
    hot = np.arange(0, I, stride)[:400]
    Q[hot] = np.abs(rng.standard_normal((len(hot), k))).astype(np.float32) + 1.0
    U[5] = -U[5]  # one row whose best items are NOT the hot ones
    lens = rng.integers(0, 200, B)
    ptr = np.zeros(B + 1, np.int64)
    np.cumsum(lens, out=ptr[1:])
    ex = np.concatenate([rng.choice(I, l, replace=False) for l in lens]).astype(np.int32)
    dU, dQ = D.to_device_padded(U, gpu), D.to_device_padded(Q, gpu)
    idx, sc = D.score_topk(dU, dQ, k, n, torch.from_numpy(ptr).to(gpu), torch.from_numpy(ex).to(gpu))
    idx, sc = idx.cpu().numpy(), sc.cpu().numpy()
    for b in range(0, B, 5):
        s, want = _oracle_topn(oracle, Q, U[b], ex[ptr[b] : ptr[b + 1]], n)
        assert np.array_equal(idx[b], want), b
        assert np.array_equal(sc[b].view(np.uint32), s[want].view(np.uint32)), b


@pytest.mark.parametrize("variant", [ROUND5, ROUND4])
def test_score_topk_fused_non_finite_scores(gpu, rng, monkeypatch, variant):
    """
    NaN and infinite scores through the fused path: its epilogue flags "not below the threshold"
    (true for NaN), stores the lane's accumulators and tests them again in the flush -- NaN must
    drop out there, +inf must be kept, -inf must lose, and rows whose threshold itself is
    infinite must still give the panel path's lists, bit for bit.
    """
    from lkpy_amd import _device as D

    _variant(monkeypatch, variant)
    monkeypatch.setenv("LK_TOPK_FUSED_MIN_USERS", "1")
    B, I, k, n = 150, 17000, 64, 50
    U = rng.standard_normal((B, k)).astype(np.float32)
    Q = rng.standard_normal((I, k)).astype(np.float32)
    Q[rng.choice(I, 300, replace=False), 3] = np.nan   # NaN score for every user
    Q[rng.choice(I, 40, replace=False), 5] = np.inf    # +inf or -inf by the sign of U[:, 5]
    Q[::16][:200, 7] = np.inf                          # many infinite scores inside the sample
    U[4] = 0.0
    U[4, 5] = 1.0   # inf * 0 = NaN elsewhere; +inf scores only
    U[B - 1, 5] = 2.0  # the last row stands in for the 106 rows past the end of its tile: their
    U[B - 1, 7] = 3.0  # +inf scores must not be written anywhere
    dU, dQ = D.to_device_padded(U, gpu), D.to_device_padded(Q, gpu)
    idx, sc = D.score_topk(dU, dQ, k, n)
    monkeypatch.setenv("LK_TOPK_FUSED_MIN_ITEMS", "1000000000")
    idx2, sc2 = D.score_topk(dU, dQ, k, n)
    assert np.array_equal(idx2.cpu().numpy(), idx.cpu().numpy())
    assert np.array_equal(sc2.cpu().numpy().view(np.uint32), sc.cpu().numpy().view(np.uint32))
    sc = sc.cpu().numpy()
    assert not np.isnan(sc[idx.cpu().numpy() >= 0]).any()


def test_score_topk_fused_in_batches(gpu, rng, monkeypatch):
    """Several batches of the fused path (LK_TOPK_FUSED_ROWS; a million-item catalogue or more than
    262 144 users in production): per-batch thresholds, candidate lists, second-tier lists and
    output offsets -- the panel path's lists, bit for bit, with and without exclusions."""
    from lkpy_amd import _device as D

    monkeypatch.setenv("LK_TOPK_FUSED_MIN_USERS", "1")
    monkeypatch.setenv("LK_TOPK_FUSED_ROWS", "8192")
    B, I, k, n = 20000, 17000, 32, 20
    U = rng.standard_normal((B, k)).astype(np.float32)
    Q = (rng.standard_normal((I, k)) * (0.2 + rng.random((I, 1)))).astype(np.float32)
    lens = rng.integers(0, 40, B)
    lens[8191] = 9000   # a heavy row at the end of the first batch
    lens[8192] = 3000   # and at the start of the second
    ptr = np.zeros(B + 1, np.int64)
    np.cumsum(lens, out=ptr[1:])
    ex = rng.integers(0, I, ptr[-1]).astype(np.int32)  # duplicates allowed, any order
    dU, dQ = D.to_device_padded(U, gpu), D.to_device_padded(Q, gpu)
    dptr, dex = torch.from_numpy(ptr).to(gpu), torch.from_numpy(ex).to(gpu)
    got = [D.score_topk(dU, dQ, k, n, dptr, dex), D.score_topk(dU, dQ, k, n)]
    monkeypatch.setenv("LK_TOPK_FUSED_MIN_ITEMS", "1000000000")  # the panel path
    want = [D.score_topk(dU, dQ, k, n, dptr, dex), D.score_topk(dU, dQ, k, n)]
    for (gi, gs), (wi, ws) in zip(got, want):
        assert torch.equal(gi, wi)
        assert torch.equal(gs.view(torch.int32), ws.view(torch.int32))


@pytest.mark.parametrize("B", [30000, 20000, 12000, 8500])
def test_score_topk_item_split_equals_unsplit(gpu, rng, monkeypatch, B):
    """Round 6: batches below one round of filter workgroups split the item tiles over 2 ... 7
    parts (30 000 / 20 000 / 12 000 / 8 500 users -> 2 / 3 / 5 / 7; interleaved tiles, per-part
    candidate sub-lists, ``cand_merge_kernel``).  Same lists and score bits as the unsplit launch
    (``LK_TOPK_SPLIT=0``), exclusion lists of every length included -- rows with more than 256
    exclusion entries take the workgroup selection tier in a split batch."""
    from lkpy_amd import _device as D

    k, n, I = 64, 100, 20000
    g = torch.Generator(device=gpu).manual_seed(B)
    U = torch.randn(B, k, device=gpu, generator=g) * 0.3
    Q = torch.randn(I, k, device=gpu, generator=g) * 0.3
    Q[7000:7100] = Q[100:200]  # exact ties between distant items (different tiles, different parts)
    U[5] = 0.0                 # every score equal: every part's sub-list overflows -> redo path
    lens = rng.integers(0, 400, B)
    lens[:4] = [0, 5000, 300, 257]
    ptr = np.zeros(B + 1, np.int64)
    np.cumsum(lens, out=ptr[1:])
    ex = rng.integers(0, I, int(ptr[-1])).astype(np.int32)  # (repeats allowed: any list is a set)
    dptr, dex = torch.from_numpy(ptr).to(gpu), torch.from_numpy(ex).to(gpu)
    idx, sc = D.score_topk(U, Q, k, n, dptr, dex)
    monkeypatch.setenv("LK_TOPK_SPLIT", "0")
    idx0, sc0 = D.score_topk(U, Q, k, n, dptr, dex)
    assert torch.equal(idx, idx0)
    assert torch.equal(sc.view(torch.int32), sc0.view(torch.int32))
    assert (idx[1] >= 0).all() and not np.isin(idx[1].cpu().numpy(), ex[ptr[1]:ptr[2]]).any()
