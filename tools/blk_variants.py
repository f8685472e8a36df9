#!/usr/bin/env python3
"""
Time the ALS epoch at k = 128 / 256 (als_blk.hip) with several builds of the library in ONE
process (variants made by tools/build_variant.py):
    python tools/blk_variants.py <k> tools/_variants/lkamd_*.so
ML-25M-shaped data; prints one JSON line per library (the default build first).
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
k = int(sys.argv[1])
ratings = synth.ml25m_like()
ui = sps.csr_array((np.full(ratings.nnz, 40.0, np.float32), ratings.indices, ratings.indptr),
                   shape=ratings.shape)
rng = np.random.default_rng(42)
Q0 = (rng.standard_normal((ui.shape[1], k), dtype=np.float32) * 0.01) ** 2
P0 = (rng.standard_normal((ui.shape[0], k), dtype=np.float32) * 0.01) ** 2
default = _native.LIB_PATH
ref = None
for path in [default] + [Path(p).resolve() for p in sys.argv[2:]]:
    _native._lib = None
    _native.LIB_PATH = Path(path)
    from lkpy_amd._als_engine import HipBackend, ImplicitALSEngine

    eng = ImplicitALSEngine(ui, k, 0.1, 0.1, P0, Q0, HipBackend(k, dev, _native.SOLVER_AUTO))
    for _ in range(2):
        eng.train_epoch()
    eng.check()
    eng.u_plan.enable_timing(True)
    eng.i_plan.enable_timing(True)
    torch.cuda.synchronize()
    n_ep = 5 if k <= 128 else 3
    t0 = time.perf_counter()
    for _ in range(n_ep):
        du, di = eng.train_epoch()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n_ep
    eng.check()
    cu, su, nu = eng.u_plan.get_timing()
    ci, si, ni = eng.i_plan.get_timing()
    Q = eng.Q.clone()
    if ref is None:
        ref = Q
    rec = {"lib": Path(path).name, "k": k, "ms_per_epoch": round(dt * 1e3, 3),
           "user_solve_ms": round(su / max(nu, 1), 3), "item_solve_ms": round(si / max(ni, 1), 3),
           "chunk_ms": round((cu + ci) / max(nu, 1), 3), "delta": float(di),
           "same_bits_as_default": bool(torch.equal(Q, ref))}
    print(json.dumps(rec), flush=True)
    del eng
    torch.cuda.empty_cache()
