#!/usr/bin/env python3
"""
Time the ALS epoch with several builds of the library (variants/lkamd_*.so, made by
recompiling als_chol.hip with different -D tunables) in ONE process:
    python tools/als_variants.py variants/*.so
"""
import json
import sys
import time
from pathlib import Path

import numpy as np
import scipy.sparse as sps
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lkpy_amd import _native, synth  # noqa: E402

dev = torch.device("cuda:0")
ratings = synth.ml25m_like()
ui = sps.csr_array((np.full(ratings.nnz, 40.0, np.float32), ratings.indices, ratings.indptr),
                   shape=ratings.shape)
k = 64
rng = np.random.default_rng(42)
Q0 = (rng.standard_normal((ui.shape[1], k), dtype=np.float32) * 0.01) ** 2
P0 = (rng.standard_normal((ui.shape[0], k), dtype=np.float32) * 0.01) ** 2
default = _native.LIB_PATH
for path in [default] + [Path(p).resolve() for p in sys.argv[1:]]:
    _native._lib = None
    _native.LIB_PATH = Path(path)
    from lkpy_amd._als_engine import HipBackend, ImplicitALSEngine

    eng = ImplicitALSEngine(ui, k, 0.1, 0.1, P0, Q0, HipBackend(k, dev, _native.SOLVER_AUTO))
    for _ in range(2):
        eng.train_epoch()
    eng.check()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            du, di = eng.train_epoch()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 10)
    eng.check()
    print(json.dumps({"lib": Path(path).name, "ms_per_epoch": round(best * 1e3, 3),
                      "epochs_per_s": round(1 / best, 1), "delta": float(di)}), flush=True)
    del eng
