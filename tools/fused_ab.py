#!/usr/bin/env python3
"""
A/B of the fused chunk + solve launch (als_fused_kernel, LK_ALS_FUSED) at k <= 64 on the
ML-25M-shaped data:  python tools/fused_ab.py [k]   -- same bits expected, times side by side.
"""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import scipy.sparse as sps
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lkpy_amd import _native, synth  # noqa: E402
from lkpy_amd._als_engine import HipBackend, ImplicitALSEngine  # noqa: E402

dev = torch.device("cuda:0")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ratings = synth.ml25m_like()
ui = sps.csr_array((np.full(ratings.nnz, 40.0, np.float32), ratings.indices, ratings.indptr),
                   shape=ratings.shape)
rng = np.random.default_rng(42)
Q0 = (rng.standard_normal((ui.shape[1], k), dtype=np.float32) * 0.01) ** 2
P0 = (rng.standard_normal((ui.shape[0], k), dtype=np.float32) * 0.01) ** 2
ref = None
for mode in ("1", "0", "1"):
    os.environ["LK_ALS_FUSED"] = mode
    eng = ImplicitALSEngine(ui, k, 0.1, 0.1, P0, Q0, HipBackend(k, dev, _native.SOLVER_AUTO))
    for _ in range(3):
        eng.train_epoch()
    eng.check()
    Q3 = eng.Q.clone()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            eng.train_epoch()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20)
    eng.check()
    eng.u_plan.enable_timing(True)
    eng.i_plan.enable_timing(True)
    for _ in range(5):
        eng.train_epoch()
    torch.cuda.synchronize()
    cu, su, nu = eng.u_plan.get_timing()
    ci, si, ni = eng.i_plan.get_timing()
    if ref is None:
        ref = Q3
    print(json.dumps({"LK_ALS_FUSED": mode, "k": k, "ms_per_epoch": round(best * 1e3, 4),
                      "user_ms": [round(cu / nu, 4), round(su / nu, 4)],
                      "item_ms": [round(ci / ni, 4), round(si / ni, 4)],
                      "same_bits_after_3_epochs": bool(torch.equal(Q3, ref))}), flush=True)
    del eng
