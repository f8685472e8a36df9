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


@pytest.mark.parametrize("k,slices", [(32, 1), (128, 1), (64, 3), (128, 2)])
def test_engine_collectives_on_rccl_world1(gpu, oracle, nccl_world1, monkeypatch, k, slices):
    """slices > 1: the overlapped half-epoch -- one plan per row slice, the in-place all-gather of
    every super-block issued with async_op=True behind its solve and waited for before the next
    half -- through RCCL (one rank: each gather is the identity).  The slice Gramians are summed
    in a different order than one full Gramian, so that case is compared at 1e-5, not bitwise."""
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

    def train(force: bool, Pi=P0, Qi=Q0, epochs=3):
        monkeypatch.setenv("LK_ALS_FORCE_COLLECTIVES", "1" if force else "0")
        monkeypatch.setenv("LK_ALS_OVERLAP_SLICES", str(slices))
        eng = ImplicitALSEngine(ui, k, 0.1, 0.1, Pi, Qi, HipBackend(k, gpu, _native.SOLVER_CHOLESKY))
        assert eng.collective == force and eng.world == 1
        assert eng.slices == (slices if force else 1) and len(eng.i_plans) == eng.slices
        for _ in range(epochs):
            du, di = eng.train_epoch()
        eng.check()
        return eng.user_embeddings(), eng.item_embeddings(), eng.otor(), float(du), float(di)

    if slices > 1:
        # (the first epochs after the tiny init are ill-conditioned -- a last-bit difference of a
        # Gramian grows to 3e-3 in three of them, measured -- so: ONE epoch from a trained state)
        Pt, Qt, _, _, _ = train(False, epochs=10)
        Pc, Qc, Gc, duc, dic = train(True, Pt, Qt, 1)
        Pp, Qp, Gp, dup, dip = train(False, Pt, Qt, 1)
    else:
        Pc, Qc, Gc, duc, dic = train(True)   # all-gather / all-reduce / broadcast through RCCL
        Pp, Qp, Gp, dup, dip = train(False)  # plain single-GPU engine
    if slices == 1:
        assert np.array_equal(Pc, Pp) and np.array_equal(Qc, Qp)
        assert np.array_equal(Gc, Gp)
        assert duc == pytest.approx(dup, rel=1e-6) and dic == pytest.approx(dip, rel=1e-6)
    else:
        rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))  # noqa: E731
        print(f"\nslices {slices}, k {k}: rel P {rel(Pc, Pp):.2e} Q {rel(Qc, Qp):.2e}")
        assert rel(Pc, Pp) < 1e-4 and rel(Qc, Qp) < 1e-4 and rel(Gc, Gp) < 1e-4
        assert duc == pytest.approx(dup, rel=1e-3) and dic == pytest.approx(dip, rel=1e-3)

    # the raw collectives the engine relies on, on HBM buffers
    full = torch.arange(4096 * 64, dtype=torch.float32, device=gpu).reshape(4096, 64)
    want = full.clone()
    dist.all_gather_into_tensor(full, full[0:4096])  # input is a view of the output
    buf = torch.ones(k * k + 1, dtype=torch.float32, device=gpu)
    dist.all_reduce(buf)
    dist.broadcast(full, src=0)
    torch.cuda.synchronize()
    assert torch.equal(full, want) and float(buf.sum()) == k * k + 1
