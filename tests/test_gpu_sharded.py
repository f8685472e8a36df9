"""
The row-sharded ALS engine's DEVICE path with more than one rank -- on ONE GPU.

Everything a rank does on the device when ``world > 1`` (``lkpy_amd/_als_engine.py``): the
relabelling with padding rows (``lk_csr_relabel`` with row_src = -1), the device transpose of the
padded matrix, plan views whose row offsets do not start at zero, half-epoch launches writing into
a row block of the replicated factor matrix, slice Gramians, the Woodbury buffers of a shard -- had
only ever run with world = 1 (the gloo tests use the host restatement with the oracle as
"kernels").  Here ``world`` ranks run as threads of this process on the same GPU and exchange
through ``LoopbackComm`` (same collectives, shared memory instead of xGMI), so only the wire is
missing.  Checked: all ranks hold bit-identical replicas, and one epoch from a trained state agrees
with the single-rank engine within 1e-4 (the slice Gramians are summed in a different order than
one full Gramian, so not bitwise).
"""
from __future__ import annotations

import threading

import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu


def _run_world(world, ui, k, P0, Q0, gpu, epochs):
    import torch

    from lkpy_amd import _native
    from lkpy_amd._als_engine import HipBackend, ImplicitALSEngine, LoopbackComm

    comms = LoopbackComm.make(world)
    out, errs = [None] * world, []

    def rank_main(r):
        try:
            torch.cuda.set_device(gpu)
            eng = ImplicitALSEngine(ui, k, 0.1, 0.1, P0, Q0,
                                    HipBackend(k, gpu, _native.SOLVER_CHOLESKY), comm=comms[r])
            assert eng.world == world and eng.rank == r and eng.collective
            assert len(eng.u_plans) == eng.slices and len(eng.i_plans) == eng.slices
            if eng.sharded_setup:  # the rank's arrays hold its own rows and nothing else
                assert eng.u_plans[0].csr.indices.numel() == eng.local_nnz[0]
                assert eng.i_plans[0].csr.indices.numel() == eng.local_nnz[1]
            for _ in range(epochs):
                du, di = eng.train_epoch()
            eng.check()
            torch.cuda.synchronize()
            out[r] = (eng.P.clone(), eng.Q.clone(), eng.user_embeddings(), eng.item_embeddings(),
                      eng.otor(), float(du), float(di), eng.u_plan.use_wb, sorted(eng._zs))
        except BaseException as e:  # noqa: BLE001 -- re-raised in the main thread
            errs.append(e)
            for c in comms:
                c.sh.barrier.abort()

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise errs[0]
    return out


def _short_row_matrix(rng, n_users, n_items, mean_len):
    lens = np.clip(rng.geometric(1.0 / mean_len, n_users), 0, n_items)
    lens[rng.random(n_users) < 0.03] = 0
    lens[:4] = [700, 300, 65, 17]  # a few long rows as well
    indptr = np.zeros(n_users + 1, np.int64)
    np.cumsum(lens, out=indptr[1:])
    indices = np.concatenate([np.sort(rng.choice(n_items, ln, replace=False)) for ln in lens])
    return sps.csr_array((np.full(indptr[-1], 40.0, np.float32), indices.astype(np.int32), indptr),
                         shape=(n_users, n_items))


@pytest.mark.parametrize("world,k,wb,slices,setup", [
    (2, 32, False, 1, "full"), (3, 64, False, 1, "full"), (2, 128, True, 1, "full"),
    (3, 256, True, 1, "full"), (3, 128, False, 1, "full"), (2, 64, False, 4, "full"),
    (3, 128, True, 3, "full"), (2, 256, True, 2, "full"),
    # LK_ALS_SETUP=sharded (the default since round 6): every rank derives only its own rows
    # (shard_local_blocks on HBM tensors: gather, searchsorted, stable sort), plans on arrays
    # that hold nnz / world entries
    (3, 64, False, 1, "sharded"), (2, 64, False, 4, "sharded"), (3, 128, True, 3, "sharded"),
    # LK_ALS_Z=sharded: Z = other @ OtOr^-1 formed once ACROSS the ranks (ShardedZ: each rank
    # its share of the rows + one all-gather) instead of once per rank
    (2, 128, True, 1, "sharded+z"), (3, 256, True, 1, "sharded+z"), (3, 128, True, 3, "sharded+z"),
    (2, 256, True, 2, "full+z")])
def test_sharded_device_path_on_one_gpu(gpu, oracle, monkeypatch, world, k, wb, slices, setup):
    import torch

    setup, _, zmode = setup.partition("+")
    monkeypatch.setenv("LK_ALS_SETUP", setup)
    monkeypatch.setenv("LK_ALS_Z", "sharded" if zmode == "z" else "replicated")

    from lkpy_amd import _native, synth
    from lkpy_amd._als_engine import HipBackend, ImplicitALSEngine

    monkeypatch.setenv("LK_ALS_WB_MIN_ROWS", "1" if wb else "0")
    # slices > 1: the overlapped half-epoch (one plan per row slice, interleaved relabelling,
    # super-block gathers, the slices of a half sharing one Z)
    monkeypatch.setenv("LK_ALS_OVERLAP_SLICES", str(slices))
    rng = np.random.default_rng(2)
    if wb:  # short rows (geometric, mean 8): the Woodbury kernels and a shard's Z buffer
        ui = _short_row_matrix(rng, 4001, 1501, 8)
    else:  # 4 876 users x 1 872 items: not multiples of 3, so padding rows exist
        ratings = synth.ml25m_like(seed=5, scale=0.03)
        ui = sps.csr_array((np.full(ratings.nnz, 40.0, np.float32), ratings.indices,
                            ratings.indptr), shape=ratings.shape)
    Q0 = oracle.als_initial_params(rng, ui.shape[1], k)
    P0 = oracle.als_initial_params(rng, ui.shape[0], k)

    # a trained state from the single-rank engine (the first epochs after the tiny init are
    # ill-conditioned: two runs that differ in the last bit of a Gramian drift apart by 5e-3 over
    # 12 epochs -- measured -- so the comparison is ONE epoch from identical, trained inputs)
    eng = ImplicitALSEngine(ui, k, 0.1, 0.1, P0, Q0, HipBackend(k, gpu, _native.SOLVER_CHOLESKY))
    for _ in range(12):
        eng.train_epoch()
    eng.check()
    Pt, Qt = eng.user_embeddings(), eng.item_embeddings()
    eng = ImplicitALSEngine(ui, k, 0.1, 0.1, Pt, Qt, HipBackend(k, gpu, _native.SOLVER_CHOLESKY))
    du, di = eng.train_epoch()
    eng.check()
    P1, Q1 = eng.user_embeddings(), eng.item_embeddings()
    assert eng.u_plan.use_wb == wb or not wb

    res = _run_world(world, ui, k, Pt, Qt, gpu, 1)
    # replicas: every rank ends with the same bits
    for r in range(1, world):
        assert torch.equal(res[r][0], res[0][0]) and torch.equal(res[r][1], res[0][1])
        assert np.array_equal(res[r][2], res[0][2]) and np.array_equal(res[r][3], res[0][3])
        assert res[r][5] == res[0][5] and res[r][6] == res[0][6]
    if wb:
        assert any(res[r][7] for r in range(world))  # some shard took the Woodbury kernels
    if zmode == "z":  # every rank took part in forming Z for the halves that use it
        assert all(res[r][8] == res[0][8] for r in range(world)) and res[0][8]

    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))  # noqa: E731
    eP, eQ = rel(res[0][2], P1), rel(res[0][3], Q1)
    print(f"\nworld {world}, k {k}, wb {wb}: sharded vs single rank, one epoch from the same state: "
          f"rel P {eP:.2e} Q {eQ:.2e}; |dP| {res[0][5]:.5f} vs {float(du):.5f}")
    assert eP < 1e-4 and eQ < 1e-4
    assert rel(res[0][4], eng.otor()) < 1e-4
    assert res[0][5] == pytest.approx(float(du), rel=1e-3)
    assert res[0][6] == pytest.approx(float(di), rel=1e-3)
    # empty rows stay zero on every rank
    empty_i = np.bincount(ui.indices, minlength=ui.shape[1]) == 0
    empty_u = np.diff(ui.indptr) == 0
    assert not res[0][3][empty_i].any() and not res[0][2][empty_u].any()
