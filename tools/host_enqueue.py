"""Host cost of enqueuing one ALS epoch (cfg2 shape at a small scale so that the GPU is never the
limit): the time train_epoch() takes to RETURN, no synchronisation inside the loop.  At N = 8 an
epoch is predicted at 0.5 ms of device time: the host must enqueue faster than
that.  python tools/host_enqueue.py [scale]"""
import cProfile
import pstats
import sys
import time
from pathlib import Path

import numpy as np
import scipy.sparse as sps

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from lkpy_amd import _native, synth  # noqa: E402
from lkpy_amd._als_engine import HipBackend, ImplicitALSEngine  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
r = synth.ml25m_like(seed=3, scale=scale)
ui = sps.csr_array((np.full(r.nnz, 40.0, np.float32), r.indices, r.indptr), shape=r.shape)
rng = np.random.default_rng(0)
k = 64
Q0 = rng.standard_normal((ui.shape[1], k), dtype=np.float32) * 0.01
P0 = rng.standard_normal((ui.shape[0], k), dtype=np.float32) * 0.01
dev = torch.device("cuda:0")
eng = ImplicitALSEngine(ui, k, 0.1, 0.1, P0 * P0, Q0 * Q0, HipBackend(k, dev, _native.SOLVER_AUTO))
for _ in range(5):
    eng.train_epoch()
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    eng.train_epoch()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"scale {scale}: host enqueue {t_host / n * 1e3:.3f} ms per epoch; with the device "
      f"{t_all / n * 1e3:.3f} ms per epoch")
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    eng.train_epoch()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
