"""cfg2 epoch and per-kernel times (plan timers: chunk kernel, solve kernels) under the summation
orders and work-unit sizes: python tools/chunk_ab.py"""
import os
import sys
import time
from pathlib import Path

import numpy as np
import scipy.sparse as sps

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from lkpy_amd import _native, synth  # noqa: E402
from lkpy_amd._als_engine import HipBackend, ImplicitALSEngine  # noqa: E402

r = synth.ml25m_like()
ui = sps.csr_array((np.full(r.nnz, 40.0, np.float32), r.indices, r.indptr), shape=r.shape)
rng = np.random.default_rng(42)
k = 64
Q0 = rng.standard_normal((ui.shape[1], k), dtype=np.float32) * 0.01
Q0 *= Q0
P0 = rng.standard_normal((ui.shape[0], k), dtype=np.float32) * 0.01
P0 *= P0
dev = torch.device("cuda:0")
for label, env in (("auto unit 1024", {}), ("auto unit 256", {"LK_ALS_REF_UNIT": "256"}),
                   ("auto unit 2048", {"LK_ALS_REF_UNIT": "2048"}),
                   ("auto unit 4096", {"LK_ALS_REF_UNIT": "4096"}),
                   ("accurate", {"LK_ALS_RHS_ORDER": "accurate"}),
                   ("auto, side stream off", {"LK_ALS_SIDE_STREAM": "0"}),
                   ("auto, ref_len 4096?", {"LK_ALS_REF_LEN": "2048"})):
    for kk in ("LK_ALS_REF_UNIT", "LK_ALS_RHS_ORDER", "LK_ALS_SIDE_STREAM", "LK_ALS_REF_LEN"):
        os.environ.pop(kk, None)
    os.environ.update(env)
    eng = ImplicitALSEngine(ui, k, 0.1, 0.1, P0, Q0, HipBackend(k, dev, _native.SOLVER_AUTO))
    for _ in range(5):
        eng.train_epoch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        eng.train_epoch()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    eng.u_plan.enable_timing(True)
    eng.i_plan.enable_timing(True)
    for _ in range(10):
        eng.train_epoch()
    torch.cuda.synchronize()
    cu, su, nu = eng.u_plan.get_timing()
    ci, si, ni = eng.i_plan.get_timing()
    print(f"{label:24s} epoch {ms:.3f} ms | user: chunk {cu / nu:.3f} solve {su / nu:.3f} | "
          f"item: chunk {ci / ni:.3f} solve {si / ni:.3f}", flush=True)
    del eng
