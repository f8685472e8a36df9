#!/usr/bin/env python3
"Explicit-feedback (biased-MF) ALS epochs on the ML-25M-shaped set: python tools/als_explicit_only.py [k]"
import json
import sys
import time
from pathlib import Path

import numpy as np
import scipy.sparse as sps
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lkpy_amd import _native, synth  # noqa: E402
from lkpy_amd._als_engine import HipBackend, ImplicitALSEngine  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
ratings = synth.ml25m_like()
# bias-normalised ratings stand-in: centre by the global mean (the bias model is host code)
vals = (ratings.data - ratings.data.mean()).astype(np.float32)
ui = sps.csr_array((vals, ratings.indices, ratings.indptr), shape=ratings.shape)
rng = np.random.default_rng(42)


def unit(n):
    m = rng.standard_normal((n, k), dtype=np.float32)
    return m / np.linalg.norm(m, axis=1, keepdims=True)


Q0, P0 = unit(ui.shape[1]), unit(ui.shape[0])
eng = ImplicitALSEngine(ui, k, 0.1, 0.1, P0, Q0, HipBackend(k, dev, _native.SOLVER_CHOLESKY),
                        explicit=True)
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
print(json.dumps({"model": "explicit (biased MF)", "k": k, "ms_per_epoch": round(best * 1e3, 3),
                  "epochs_per_s": round(1 / best, 1), "deltas": [float(du), float(di)]}))
