#!/usr/bin/env python3
"ALS with the CG solver on the ML-25M-shaped set: python tools/cg_only.py [k] [tol]"
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

k = int(sys.argv[1]) if len(sys.argv) > 1 else 128
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-6
dev = torch.device("cuda:0")
ratings = synth.ml25m_like()
ui = sps.csr_array((np.full(ratings.nnz, 40.0, np.float32), ratings.indices, ratings.indptr),
                   shape=ratings.shape)
rng = np.random.default_rng(42)
Q0 = (rng.standard_normal((ui.shape[1], k), dtype=np.float32) * 0.01) ** 2
P0 = (rng.standard_normal((ui.shape[0], k), dtype=np.float32) * 0.01) ** 2
eng = ImplicitALSEngine(ui, k, 0.1, 0.1, P0, Q0, HipBackend(k, dev, _native.SOLVER_CG))
eng.u_plan.set_cg(tol, 0)
eng.i_plan.set_cg(tol, 0)
eng.train_epoch()
eng.check()
torch.cuda.synchronize()
times = []
for _ in range(4):
    t0 = time.perf_counter()
    du, di = eng.train_epoch()
    torch.cuda.synchronize()
    times.append(time.perf_counter() - t0)
eng.check()
iu_, ru_ = eng.u_plan.cg_stats()
ii_, ri_ = eng.i_plan.cg_stats()
print(json.dumps({"k": k, "tol": tol, "ms_per_epoch": [round(t * 1e3, 2) for t in times],
                  "deltas": [float(du), float(di)],
                  "cg_iterations_per_row": {"user": round(iu_ / max(ru_, 1), 2),
                                            "item": round(ii_ / max(ri_, 1), 2)},
                  "rows": [ru_, ri_]}))
