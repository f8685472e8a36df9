#!/usr/bin/env python3
"""
Secondary measurements quoted in DESIGN.md (not the headline bench): other BASELINE.json
configs run on ONE MI355X -- ALS with the CG solver at k = 128 (cfg4's shape), dense top-K
over all users x all items, batched item-kNN scoring, save_nbrs truncation.
    python tools/measure_extra.py            (through gpurun)
"""
import json
import sys
import time
from pathlib import Path

import numpy as np
import scipy.sparse as sps
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lkpy_amd import _device as D  # noqa: E402
from lkpy_amd import _knn_bench, _native, synth  # noqa: E402
from lkpy_amd._als_engine import HipBackend, ImplicitALSEngine  # noqa: E402


def sync():
    torch.cuda.synchronize()


def main():
    dev = torch.device("cuda:0")
    out = {}
    ratings = synth.ml25m_like()
    ui = sps.csr_array((np.full(ratings.nnz, 40.0, np.float32), ratings.indices,
                        ratings.indptr), shape=ratings.shape)
    rng = np.random.default_rng(42)

    # ---- ALS k = 128, CG solver (cfg4 on one GPU) ----
    for k, iters, tol in ((128, 0, 1e-6),):
        Q0 = (rng.standard_normal((ui.shape[1], k), dtype=np.float32) * 0.01) ** 2
        P0 = (rng.standard_normal((ui.shape[0], k), dtype=np.float32) * 0.01) ** 2
        eng = ImplicitALSEngine(ui, k, 0.1, 0.1, P0, Q0, HipBackend(k, dev, _native.SOLVER_CG))
        eng.u_plan.set_cg(tol, iters)
        eng.i_plan.set_cg(tol, iters)
        eng.train_epoch()
        eng.check()
        sync()
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            du, di = eng.train_epoch()
        sync()
        dt = (time.perf_counter() - t0) / n
        eng.check()
        out[f"als_cg_k{k}"] = {"ms_per_epoch": round(dt * 1e3, 2), "epochs_per_s": round(1 / dt, 3),
                               "tol": tol, "deltas": [float(du), float(di)]}
        P = eng.P
        Q = eng.Q
        # ---- dense top-K over ALL users x ALL items with these factors ----
        ptr = torch.from_numpy(eng.u_plan.csr.h_indptr.astype(np.int64)).to(dev)
        sync()
        t0 = time.perf_counter()
        idx, sc = D.score_topk(P, Q, k, 100, ptr, eng.u_plan.csr.indices)
        sync()
        dt = time.perf_counter() - t0
        B, I = P.shape[0], Q.shape[0]
        out[f"topk_k{k}"] = {"users": B, "items": I, "n": 100, "seconds": round(dt, 4),
                             "users_per_s": round(B / dt, 1),
                             "tflops": round(2.0 * B * I * k / dt / 1e12, 2)}
        del eng, P, Q, idx, sc

    # ---- item-kNN: build with save_nbrs, batched scoring ----
    uin, iun, means = _knn_bench.prepare_explicit(ratings)
    dui, diu = D.DeviceCSR.from_scipy(uin, dev), D.DeviceCSR.from_scipy(iun, dev)
    sync()
    t0 = time.perf_counter()
    sims = D.iknn_build(dui, diu, 1e-6, 100)
    sync()
    out["knn_build_save_nbrs_100"] = {"seconds": round(time.perf_counter() - t0, 4),
                                      "nnz": int(sims.indices.shape[0])}
    csr = sps.csr_array(ratings)
    users = rng.choice(csr.shape[0], 10000, replace=False)
    r_ptr = np.zeros(len(users) + 1, np.int64)
    np.cumsum(np.diff(csr.indptr)[users], out=r_ptr[1:])
    r_idx = np.concatenate([csr.indices[csr.indptr[u]:csr.indptr[u + 1]] for u in users])
    r_val = np.concatenate([csr.data[csr.indptr[u]:csr.indptr[u + 1]] for u in users])
    r_val = (r_val - means[r_idx]).astype(np.float32)
    tgt = np.sort(rng.choice(csr.shape[1], 100, replace=False)).astype(np.int32)
    t_ptr = np.arange(len(users) + 1, dtype=np.int64) * 100
    t_idx = np.tile(tgt, len(users))
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    args = (sims, to(r_ptr), to(r_idx.astype(np.int32)), to(r_val), to(t_ptr), to(t_idx), 100, 1)
    D.iknn_score_batch(*args)
    sync()
    t0 = time.perf_counter()
    s, c = D.iknn_score_batch(*args)
    sync()
    dt = time.perf_counter() - t0
    out["knn_score_batch"] = {"queries": len(users), "targets_per_query": 100,
                              "seconds": round(dt, 4), "queries_per_s": round(len(users) / dt, 1),
                              "scored": int(torch.isfinite(s).sum())}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
