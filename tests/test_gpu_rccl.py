"""
The RCCL path executed on DEVICE tensors (one GPU, world size 1, backend ``nccl`` = RCCL).

The ALS engine's collectives -- the in-place ``all_gather_into_tensor`` whose input is a slice
view of its output, the single ``k*k + 1``-float all-reduce carrying the slice Gramian and the
squared delta, and the broadcast of the initial factors (``lkpy_amd/_als_engine.py``; SURVEY.md
section 8e) -- only run when there is more than one rank.  ``LK_ALS_FORCE_COLLECTIVES=1`` makes
a single rank issue them too, so that every call has gone through RCCL with HBM buffers at least
once before an 8-GPU node sees the code.  With one rank each collective is the identity, so the
trained factors must be BIT-IDENTICAL to the plain engine's.  (world > 1 semantics: the gloo
tests in ``tests/test_distributed_cpu.py``.)
"""
from __future__ import annotations

import os
import socket

import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture()
def nccl_world1(gpu, monkeypatch):
    import torch
    import torch.distributed as dist

    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", str(_free_port()))
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    torch.cuda.set_device(gpu)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=gpu)
    try:
        yield dist
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("k", [32, 128])
def test_engine_collectives_on_rccl_world1(gpu, oracle, nccl_world1, monkeypatch, k):
    import torch

    from lkpy_amd import _native, synth
    from lkpy_amd._als_engine import HipBackend, ImplicitALSEngine

    dist = nccl_world1
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    ratings = synth.ml25m_like(seed=11, scale=0.02)
    ui = sps.csr_array((np.full(ratings.nnz, 40.0, np.float32), ratings.indices, ratings.indptr),
                       shape=ratings.shape)
    rng = np.random.default_rng(1)
    Q0 = oracle.als_initial_params(rng, ui.shape[1], k)
    P0 = oracle.als_initial_params(rng, ui.shape[0], k)

    def train(force: bool):
        monkeypatch.setenv("LK_ALS_FORCE_COLLECTIVES", "1" if force else "0")
        eng = ImplicitALSEngine(ui, k, 0.1, 0.1, P0, Q0, HipBackend(k, gpu, _native.SOLVER_CHOLESKY))
        assert eng.collective == force and eng.world == 1
        for _ in range(3):
            du, di = eng.train_epoch()
        eng.check()
        return eng.user_embeddings(), eng.item_embeddings(), eng.otor(), float(du), float(di)

    Pc, Qc, Gc, duc, dic = train(True)   # all-gather / all-reduce / broadcast through RCCL
    Pp, Qp, Gp, dup, dip = train(False)  # plain single-GPU engine
    assert np.array_equal(Pc, Pp) and np.array_equal(Qc, Qp)
    assert np.array_equal(Gc, Gp)
    assert duc == pytest.approx(dup, rel=1e-6) and dic == pytest.approx(dip, rel=1e-6)

    # the raw collectives the engine relies on, on HBM buffers
    full = torch.arange(4096 * 64, dtype=torch.float32, device=gpu).reshape(4096, 64)
    want = full.clone()
    dist.all_gather_into_tensor(full, full[0:4096])  # input is a view of the output
    buf = torch.ones(k * k + 1, dtype=torch.float32, device=gpu)
    dist.all_reduce(buf)
    dist.broadcast(full, src=0)
    torch.cuda.synchronize()
    assert torch.equal(full, want) and float(buf.sum()) == k * k + 1
