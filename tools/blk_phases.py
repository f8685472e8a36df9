#!/usr/bin/env python3
"""
Per-phase shader-clock cycles of the k = 128 / 256 row solve (als_blk.hip built with
-DLK_BLK_PHASES, tools/build_variant.py):  python tools/blk_phases.py <k> <variant.so>
"""
import ctypes
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
_native._lib = None
_native.LIB_PATH = Path(sys.argv[2]).resolve()
from lkpy_amd._als_engine import HipBackend, ImplicitALSEngine  # noqa: E402

ratings = synth.ml25m_like()
ui = sps.csr_array((np.full(ratings.nnz, 40.0, np.float32), ratings.indices, ratings.indptr),
                   shape=ratings.shape)
rng = np.random.default_rng(42)
Q0 = (rng.standard_normal((ui.shape[1], k), dtype=np.float32) * 0.01) ** 2
P0 = (rng.standard_normal((ui.shape[0], k), dtype=np.float32) * 0.01) ** 2
eng = ImplicitALSEngine(ui, k, 0.1, 0.1, P0, Q0, HipBackend(k, dev, _native.SOLVER_AUTO))
for _ in range(2):
    eng.train_epoch()
eng.check()
lib = _native.load()
names = ["gram", "publish+bar", "diag+panel", "writeback+bar", "fwd+update", "back", "len", "whole"]
for half in ("user", "item"):
    n = eng.P.shape[0] if half == "user" else eng.Q.shape[0]
    buf = torch.zeros((n, 8), dtype=torch.int32, device=dev)
    lib.lk_blk_phase_set(ctypes.c_void_p(buf.data_ptr()))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if half == "user":
        eng.backend.half_epoch(eng.u_plan, eng.P, eng.Q, eng._qtq)
    else:
        eng.backend.half_epoch(eng.i_plan, eng.Q, eng.P, eng.backend.gramian(eng.P, eng.item_reg))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    lib.lk_blk_phase_set(ctypes.c_void_p(0))
    b = buf.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    b = b[b[:, 7] > 0]
    out = {"half": half, "k": k, "rows": int(len(b)), "ms": round(ms, 3)}
    for tag, sel in [("all", slice(None)), ("len>2048", b[:, 6] > 2048),
                     ("512..2048", (b[:, 6] > 512) & (b[:, 6] <= 2048)),
                     ("128..512", (b[:, 6] > 128) & (b[:, 6] <= 512)), ("<=128", b[:, 6] <= 128)]:
        x = b[sel]
        if len(x):
            out[tag] = {"n": int(len(x)), **{nm: int(x[:, i].mean()) for i, nm in enumerate(names)}}
    print(json.dumps(out), flush=True)
