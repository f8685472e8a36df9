#!/usr/bin/env python3
"""
Per-phase shader-clock cycles of the item-kNN recommend kernel (iknn_recommend.hip built with
-DLK_REC_PHASES, tools/build_variant.py):  python tools/knnrec_phases.py <variant.so>
"""
import ctypes
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lkpy_amd import _native, synth  # noqa: E402

_native._lib = None
_native.LIB_PATH = Path(sys.argv[1]).resolve()
from lkpy_amd import _device as D  # noqa: E402
from lkpy_amd import _knn_bench  # noqa: E402

dev = torch.device("cuda:0")
ratings = synth.ml25m_like()
dui, diu, means, _ = D.iknn_prepare(ratings, True, dev)
sims = D.iknn_build(dui, diu, 1.0e-6, 100)
del dui, diu
lib = _native.load()
buf = torch.zeros(8, dtype=torch.int64, device=dev)
lib.lk_rec_phase_set(ctypes.c_void_p(buf.data_ptr()))
res = _knn_bench._recommend_leg(D, ratings, means, sims, dev)
torch.cuda.synchronize()
b = buf.cpu().numpy()
names = ["setup", "zero", "count", "scan+region", "fill", "score", "copyout"]
tot = float(b[:7].sum())
print(json.dumps({"seconds": res["seconds"], "tasks": int(b[7]),
                  "cycles_per_task": {n: round(float(b[i]) / max(int(b[7]), 1), 1)
                                      for i, n in enumerate(names)},
                  "share": {n: round(float(b[i]) / tot, 3) for i, n in enumerate(names)}}))
