"""
CPU, world_size 2, gloo: the row-sharding / exchange logic of ImplicitALSEngine
(relabelling, per-rank row blocks, in-place all-gather, Gramian and delta all-reduce).
The HIP kernels cannot run here, so the ORACLE stands in for the arithmetic through the
engine's backend interface -- test infrastructure only; the product backend is HipBackend.
"""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sps
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class OracleBackend:
    def __init__(self, k):
        from oracle import lk_oracle

        self.lko = lk_oracle
        self.k = self.kp = k

    def make_plan(self, local_csr):
        m = sps.csr_array(local_csr)
        m.sort_indices()
        return m

    def upload(self, mat):
        return torch.from_numpy(np.ascontiguousarray(mat, dtype=np.float32).copy())

    def download(self, t):
        return t.numpy().copy()

    def gramian(self, rows, reg):
        r = rows.numpy().astype(np.float64)
        return torch.from_numpy((r.T @ r + reg * np.eye(self.k)).astype(np.float32))

    def half_epoch(self, plan, this_slice, other_full, otor):
        view = this_slice.numpy()  # shares memory: updated in place like the kernel does
        d = self.lko.als_half_epoch(plan, view, other_full.numpy(), otor.numpy(), 2)
        return torch.tensor([d], dtype=torch.float32)

    def half_epoch_explicit(self, plan, this_slice, other_full, reg):
        view = this_slice.numpy()
        d = self.lko.als_explicit_half_epoch(plan, view, other_full.numpy(), reg, 2)
        return torch.tensor([d], dtype=torch.float32)

    def check(self, plan):
        pass


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ui, k, P0, Q0, epochs, out, explicit=False, slices=0,
            setup="full"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LK_ALS_OVERLAP_SLICES"] = str(slices)
    os.environ["LK_ALS_SETUP"] = setup
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lkpy_amd._als_engine import ImplicitALSEngine

        eng = ImplicitALSEngine(ui, k, 0.1, 0.2, P0, Q0, OracleBackend(k), explicit=explicit)
        assert eng.sharded_setup == (setup in ("", "sharded"))
        assert eng.slices == (slices if slices > 0 else 1) and len(eng.u_plans) == eng.slices
        deltas = []
        for _ in range(epochs):
            du, di = eng.train_epoch()
            deltas.append((float(du), float(di)))
        out[rank] = (eng.user_embeddings(), eng.item_embeddings(),
                     None if explicit else eng.otor(), deltas, eng.local_nnz)
    finally:
        dist.destroy_process_group()


def test_deal_rows_balances_and_round_trips():
    from lkpy_amd._als_engine import deal_rows

    rng = np.random.default_rng(0)
    lens = rng.geometric(0.02, 1001)
    for world in (1, 2, 3, 8):
        new_of_old, old_of_new, rpr = deal_rows(lens, world)
        assert rpr * world >= len(lens) and len(old_of_new) == rpr * world
        assert np.array_equal(np.sort(new_of_old), np.sort(np.flatnonzero(old_of_new >= 0)))
        assert np.array_equal(old_of_new[new_of_old], np.arange(len(lens)))
        per_rank = [lens[old_of_new[r * rpr:(r + 1) * rpr][old_of_new[r * rpr:(r + 1) * rpr] >= 0]].sum()
                    for r in range(world)]
        assert max(per_rank) - min(per_rank) <= lens.max()  # nnz-balanced to one row
        # the sliced layout of the overlapped half-epoch: slice s of ALL ranks is contiguous, a
        # (rank, slice) block holds every slices-th row dealt to the rank
        for S in (2, 3, 4):
            new2, old2, rpr2 = deal_rows(lens, world, S)
            m = rpr2 // S
            assert rpr2 % S == 0 and rpr2 >= rpr and len(old2) == world * rpr2
            assert np.array_equal(old2[new2], np.arange(len(lens)))
            assert np.array_equal(np.sort(new2), np.sort(np.flatnonzero(old2 >= 0)))
            for r in range(world):
                mine = np.concatenate([old2[s_ * world * m + r * m: s_ * world * m + (r + 1) * m]
                                       for s_ in range(S)])
                base = old_of_new[r * rpr:(r + 1) * rpr]
                assert np.array_equal(np.sort(mine[mine >= 0]), np.sort(base[base >= 0]))
                blk = [lens[b[b >= 0]].sum() for b in
                       (old2[s_ * world * m + r * m: s_ * world * m + (r + 1) * m] for s_ in range(S))]
                assert max(blk) - min(blk) <= 2 * lens.max() + lens.sum() // (world * 50)


@pytest.mark.parametrize("world,slices,setup", [(2, 0, "full"), (2, 3, "full"),
                                                (2, 0, "sharded"), (2, 3, "sharded"), (2, 0, "")])
def test_sharded_engine_matches_single_process(oracle, world, slices, setup):
    """(``setup = "sharded"``, and ``""`` = the variable unset, the default since round 6: every
    rank cuts only its own rows out of the original matrix, ``shard_local_blocks``; same factors,
    same per-rank entry counts.  ``"full"``: rounds 1-5's set-up, LK_ALS_SETUP=full)"""
    rng = np.random.default_rng(5)
    n_users, n_items, k, epochs = 301, 157, 8, 3
    dense = rng.random((n_users, n_items)) < 0.06
    dense[:, 5] = False  # an item nobody rated
    dense[7, :] = False  # a user without interactions
    ui = sps.csr_array(dense.astype(np.float32) * 40.0)
    ui.eliminate_zeros()
    Q0 = oracle.als_initial_params(rng, n_items, k)
    P0 = oracle.als_initial_params(rng, n_users, k)

    # single process, plain reference order (no relabelling)
    P, Q = P0.copy(), Q0.copy()
    iu = sps.csr_array(ui.T)
    iu.sort_indices()
    ref_d = []
    for _ in range(epochs):
        du = oracle.als_half_epoch(ui, P, Q, oracle.implicit_otor(Q, 0.1))
        di = oracle.als_half_epoch(iu, Q, P, oracle.implicit_otor(P, 0.2))
        ref_d.append((du, di))

    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    # (slices = 3: the half-epochs run slice by slice with asynchronous gloo all-gathers of the
    # interleaved super-blocks)
    mp.spawn(_worker, args=(world, port, ui, k, P0, Q0, epochs, out, False, slices, setup),
             nprocs=world, join=True)
    assert sorted(out.keys()) == list(range(world))
    for r in range(world):
        gP, gQ, gO, gd, lnnz = out[r]
        assert np.allclose(gP, P, rtol=1e-3, atol=1e-5) and np.allclose(gQ, Q, rtol=1e-3, atol=1e-5)
        assert np.all(gQ[5] == 0) and np.all(gP[7] == 0)
        assert np.allclose(gO, oracle.implicit_otor(Q, 0.1), rtol=1e-3, atol=1e-5)
        assert np.allclose(np.array(gd), np.array(ref_d), rtol=1e-3)
    # every rank ends with identical replicas
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    # and the shards really split the work
    assert out[0][4][0] + out[1][4][0] == ui.nnz and abs(out[0][4][0] - out[1][4][0]) < 0.2 * ui.nnz


@pytest.mark.parametrize("world,slices", [(2, 0), (2, 2)])
def test_sharded_explicit_engine_matches_single_process(oracle, world, slices):
    "The biased-MF (explicit) mode of the engine: same sharding and exchanges, no Gramian."
    rng = np.random.default_rng(6)
    n_users, n_items, k, epochs = 211, 97, 6, 3
    mask = rng.random((n_users, n_items)) < 0.08
    mask[:, 3] = False
    mask[11, :] = False
    vals = rng.standard_normal((n_users, n_items)).astype(np.float32)
    vals[vals == 0] = 0.5
    ui = sps.csr_array(np.where(mask, vals, 0).astype(np.float32))
    ui.eliminate_zeros()
    Q0 = oracle.als_explicit_initial_params(rng, n_items, k)
    P0 = oracle.als_explicit_initial_params(rng, n_users, k)
    P, Q = P0.copy(), Q0.copy()
    iu = sps.csr_array(ui.T)
    iu.sort_indices()
    ref_d = []
    for _ in range(epochs):
        du = oracle.als_explicit_half_epoch(ui, P, Q, 0.1)
        di = oracle.als_explicit_half_epoch(iu, Q, P, 0.2)
        ref_d.append((du, di))
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ui, k, P0, Q0, epochs, out, True, slices),
             nprocs=world, join=True)
    for r in range(world):
        gP, gQ, _, gd, _ = out[r]
        assert np.allclose(gP, P, rtol=1e-3, atol=1e-5) and np.allclose(gQ, Q, rtol=1e-3, atol=1e-5)
        assert np.all(gQ[3] == 0) and np.all(gP[11] == 0)
        assert np.allclose(np.array(gd), np.array(ref_d), rtol=1e-3)
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize("explicit", [False, True])
def test_engine_single_process_relabelling_round_trip(oracle, explicit):
    """world = 1 (no process group): the longest-first relabelling of users and items is
    invisible from outside -- factors come back in the caller's order and equal the plain
    reference-order run."""
    from lkpy_amd._als_engine import ImplicitALSEngine

    rng = np.random.default_rng(9)
    n_users, n_items, k = 120, 70, 5
    mask = rng.random((n_users, n_items)) < 0.1
    vals = np.where(mask, rng.standard_normal((n_users, n_items)) if explicit else 40.0, 0)
    ui = sps.csr_array(vals.astype(np.float32))
    ui.eliminate_zeros()
    iu = sps.csr_array(ui.T)
    iu.sort_indices()
    init = oracle.als_explicit_initial_params if explicit else oracle.als_initial_params
    Q0, P0 = init(rng, n_items, k), init(rng, n_users, k)
    P, Q = P0.copy(), Q0.copy()
    for _ in range(2):
        if explicit:
            oracle.als_explicit_half_epoch(ui, P, Q, 0.1)
            oracle.als_explicit_half_epoch(iu, Q, P, 0.2)
        else:
            oracle.als_half_epoch(ui, P, Q, oracle.implicit_otor(Q, 0.1))
            oracle.als_half_epoch(iu, Q, P, oracle.implicit_otor(P, 0.2))
    eng = ImplicitALSEngine(ui, k, 0.1, 0.2, P0, Q0, OracleBackend(k), explicit=explicit)
    assert eng.world == 1
    for _ in range(2):
        eng.train_epoch()
    assert np.allclose(eng.user_embeddings(), P, rtol=1e-3, atol=1e-5)
    assert np.allclose(eng.item_embeddings(), Q, rtol=1e-3, atol=1e-5)
    if not explicit:
        assert np.allclose(eng.otor(), oracle.implicit_otor(Q, 0.1), rtol=1e-3, atol=1e-5)


def test_engine_deferred_initial_factors(oracle):
    """
    ``defer_init=True`` + ``set_initial`` (the trainer draws the host random numbers while the
    engine uploads / relabels the matrix) is the same engine as passing the factors up front.
    """
    from lkpy_amd._als_engine import ImplicitALSEngine

    rng = np.random.default_rng(4)
    n_users, n_items, k = 90, 50, 4
    ui = sps.csr_array(np.where(rng.random((n_users, n_items)) < 0.12, 40.0, 0).astype(np.float32))
    ui.eliminate_zeros()
    Q0, P0 = oracle.als_initial_params(rng, n_items, k), oracle.als_initial_params(rng, n_users, k)
    a = ImplicitALSEngine(ui, k, 0.1, 0.2, P0, Q0, OracleBackend(k))
    b = ImplicitALSEngine(ui, k, 0.1, 0.2, None, None, OracleBackend(k), defer_init=True)
    assert b._qtq is None
    b.set_initial(P0, Q0)
    for eng in (a, b):
        eng.train_epoch()
    assert np.array_equal(a.user_embeddings(), b.user_embeddings())
    assert np.array_equal(a.item_embeddings(), b.item_embeddings())
    assert np.array_equal(np.asarray(a.otor()), np.asarray(b.otor()))


# ---------------------------------------------------------------------------------------
# item-kNN build / dense top-N sharded over ranks (lkpy_amd/_sharded.py), oracle standing in
# ---------------------------------------------------------------------------------------


def _knn_worker(rank, world, port, ui, iu, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lkpy_amd import _sharded
        from oracle import lk_oracle as lko

        w = _sharded.iknn_row_weights(ui.indptr, iu.indptr, iu.indices)

        def build(lo, hi):  # the oracle's sim_row for this rank's rows
            blk = lko.iknn_build_rows(ui, iu, np.arange(lo, hi, dtype=np.int32), 1.0e-6, 20)
            return (torch.from_numpy(blk.indptr.astype(np.int64)),
                    torch.from_numpy(blk.indices.astype(np.int32)),
                    torch.from_numpy(blk.data.astype(np.float32)))

        ranges, res = _sharded.build_rows_sharded(build, iu.shape[0], w)
        out[rank] = (ranges, None if res is None else tuple(t.numpy() for t in res))
    finally:
        dist.destroy_process_group()


def test_knn_build_sharded_world2(oracle, ml_small):
    ui, iu, _m, _ = oracle.iknn_prepare(ml_small["rmat"], True)
    want = oracle.iknn_build(ui, iu, 1.0e-6, 20)
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_knn_worker, args=(2, _free_port(), ui, iu, out), nprocs=2, join=True)
    ranges, res = out[0]
    assert out[1][1] is None  # only the destination rank holds the stitched matrix
    assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] == ui.shape[1]
    # balanced by multiply-accumulates, not by row count
    from lkpy_amd import _sharded

    w = _sharded.iknn_row_weights(ui.indptr, iu.indptr, iu.indices)
    a, b = w[: ranges[0][1]].sum(), w[ranges[0][1] :].sum()
    assert abs(a - b) / (a + b) < 0.05
    ptr, idx, val = res
    assert ptr.dtype == np.int64 and np.array_equal(ptr, want.indptr)
    assert np.array_equal(idx, want.indices)
    assert np.array_equal(val.view(np.uint32), want.data.view(np.uint32))


def _topk_worker(rank, world, port, P, Q, n, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lkpy_amd import _sharded
        from oracle import lk_oracle as lko

        def block(lo, hi):
            idx = np.full((hi - lo, n), -1, np.int32)
            sc = np.full((hi - lo, n), np.nan, np.float32)
            for u in range(lo, hi):
                s = lko.score_dense(Q, P[u])
                top = lko.argtopn(s, n)
                idx[u - lo, : len(top)] = top
                sc[u - lo, : len(top)] = s[top]
            return torch.from_numpy(idx), torch.from_numpy(sc)

        ranges, res = _sharded.topk_sharded(block, P.shape[0], n)
        out[rank] = (ranges, None if res is None else tuple(t.numpy() for t in res))
    finally:
        dist.destroy_process_group()


def test_topk_sharded_world2(oracle, rng):
    P = rng.standard_normal((37, 16)).astype(np.float32)  # odd count: uneven blocks
    Q = rng.standard_normal((300, 16)).astype(np.float32)
    n = 10
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_topk_worker, args=(2, _free_port(), P, Q, n, out), nprocs=2, join=True)
    ranges, res = out[0]
    assert ranges == [(0, 19), (19, 37)] and out[1][1] is None
    idx, sc = res
    for u in range(37):
        s = oracle.score_dense(Q, P[u])
        assert np.array_equal(idx[u], oracle.argtopn(s, n))
        assert np.array_equal(sc[u], s[idx[u]])


@pytest.mark.parametrize("world,S", [(2, 1), (3, 2), (2, 3)])
def test_shard_local_blocks_are_the_rows_of_the_full_relabelling(world, S):
    """The per-rank set-up (``shard_local_blocks``, plain torch: the product runs it on HBM
    tensors): a rank's user rows = the original rows in dealt order with mapped columns and the
    entry order kept; its item rows = the columns of the original matrix with the entries by
    ascending ORIGINAL user (the reference's order) or ascending new user -- entry for entry what
    the full relabel + stable transpose lists for those rows; padding rows are empty."""
    from lkpy_amd._als_engine import deal_rows, shard_local_blocks

    rng = np.random.default_rng(3)
    n_users, n_items = 301, 157
    dense = rng.random((n_users, n_items)) < 0.06
    dense[:, 5] = False
    dense[7, :] = False
    ui = sps.csr_array(dense * rng.random((n_users, n_items)).astype(np.float32))
    ui.eliminate_zeros()
    ui.sort_indices()
    csc = sps.csc_array(ui)
    u_new, u_old, u_rpr = deal_rows(np.diff(ui.indptr), world, S)
    i_new, i_old, i_rpr = deal_rows(np.bincount(ui.indices, minlength=n_items), world, S)
    um, im = u_rpr // S, i_rpr // S
    seen_u = seen_i = 0
    for r in range(world):
        ub = [(s_ * world * um + r * um, s_ * world * um + (r + 1) * um) for s_ in range(S)]
        ib = [(s_ * world * im + r * im, s_ * world * im + (r + 1) * im) for s_ in range(S)]
        for by_new in (True, False):
            out = shard_local_blocks(torch.from_numpy(ui.indptr.astype(np.int64)),
                                     torch.from_numpy(ui.indices.astype(np.int32)),
                                     torch.from_numpy(ui.data), u_old, u_new, i_new, ub, ib, by_new)
            hp, _, idx, val = out["u"]
            for loc, rn in enumerate(np.concatenate([np.arange(lo, hi) for lo, hi in ub])):
                o = u_old[rn]
                gi, gv = idx[hp[loc]:hp[loc + 1]].numpy(), val[hp[loc]:hp[loc + 1]].numpy()
                if o < 0:
                    assert len(gi) == 0
                    continue
                seg = slice(ui.indptr[o], ui.indptr[o + 1])
                assert np.array_equal(gi, i_new[ui.indices[seg]]) and np.array_equal(gv, ui.data[seg])
            seen_u += int(hp[-1]) if by_new else 0
            hp, _, idx, val = out["i"]
            for loc, rn in enumerate(np.concatenate([np.arange(lo, hi) for lo, hi in ib])):
                o = i_old[rn]
                gi, gv = idx[hp[loc]:hp[loc + 1]].numpy(), val[hp[loc]:hp[loc + 1]].numpy()
                if o < 0:
                    assert len(gi) == 0
                    continue
                seg = slice(csc.indptr[o], csc.indptr[o + 1])  # ascending original user
                wi, wv = u_new[csc.indices[seg]], csc.data[seg]
                if by_new:
                    p = np.argsort(wi, kind="stable")
                    wi, wv = wi[p], wv[p]
                assert np.array_equal(gi, wi) and np.array_equal(gv, wv)
            seen_i += int(hp[-1]) if by_new else 0
    assert seen_u == ui.nnz and seen_i == ui.nnz  # the ranks' rows partition the matrix
    # the exclusion lists of a top-N call over all users (bench's sharded leg under this set-up)
    from lkpy_amd._als_engine import relabelled_user_lists

    ptr, ex = relabelled_user_lists(ui, u_old, i_new)
    assert ptr[-1] == ui.nnz and len(ptr) == len(u_old) + 1
    for rn in range(0, len(u_old), 7):
        o = u_old[rn]
        want = i_new[ui.indices[ui.indptr[o]:ui.indptr[o + 1]]] if o >= 0 else np.zeros(0, np.int64)
        assert np.array_equal(ex[ptr[rn]:ptr[rn + 1]], want)


def test_balanced_ranges_edge_cases():
    from lkpy_amd._sharded import balanced_ranges, shard_rows_even

    assert balanced_ranges(np.array([1.0, 1, 1, 1]), 2) == [(0, 2), (2, 4)]
    r = balanced_ranges(np.array([100.0, 1, 1, 1, 1]), 3)
    assert r[0][0] == 0 and r[-1][1] == 5 and all(a[1] == b[0] for a, b in zip(r, r[1:]))
    assert balanced_ranges(np.zeros(0), 2) == [(0, 0), (0, 0)]
    assert shard_rows_even(5, 2) == [(0, 3), (3, 5)] and shard_rows_even(2, 4)[-1] == (2, 2)


@pytest.mark.parametrize("world", [2, 3])
def test_loopback_comm_equals_gloo_semantics(oracle, world):
    """``LoopbackComm`` (ranks as threads of one process; used on the GPU box to run the sharded
    DEVICE path with several ranks on one GPU, tests/test_gpu_sharded.py) gives the engine the same
    results as the single-process reference order -- the same check the gloo test above makes."""
    import threading

    from lkpy_amd._als_engine import ImplicitALSEngine, LoopbackComm

    rng = np.random.default_rng(6)
    n_users, n_items, k, epochs = 211, 97, 8, 3
    dense = rng.random((n_users, n_items)) < 0.08
    dense[:, 3] = False
    ui = sps.csr_array(dense.astype(np.float32) * 40.0)
    ui.eliminate_zeros()
    Q0 = oracle.als_initial_params(rng, n_items, k)
    P0 = oracle.als_initial_params(rng, n_users, k)
    P, Q = P0.copy(), Q0.copy()
    iu = sps.csr_array(ui.T)
    iu.sort_indices()
    for _ in range(epochs):
        oracle.als_half_epoch(ui, P, Q, oracle.implicit_otor(Q, 0.1))
        oracle.als_half_epoch(iu, Q, P, oracle.implicit_otor(P, 0.2))

    comms = LoopbackComm.make(world)
    out, errs = [None] * world, []

    def rank_main(r):
        try:
            eng = ImplicitALSEngine(ui, k, 0.1, 0.2, P0, Q0, OracleBackend(k), comm=comms[r])
            assert eng.world == world and eng.rank == r
            for _ in range(epochs):
                eng.train_epoch()
            out[r] = (eng.user_embeddings(), eng.item_embeddings())
        except BaseException as e:  # noqa: BLE001
            errs.append(e)
            for c in comms:
                c.sh.barrier.abort()

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for r in range(world):
        assert np.array_equal(out[r][0], out[0][0]) and np.array_equal(out[r][1], out[0][1])
    assert np.linalg.norm(out[0][0] - P) / np.linalg.norm(P) < 1e-4
    assert np.linalg.norm(out[0][1] - Q) / np.linalg.norm(Q) < 1e-4
