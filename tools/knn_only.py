#!/usr/bin/env python3
"Item-kNN build timing only (ML-25M-shaped synthetic): python tools/knn_only.py"
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lkpy_amd import _knn_bench, synth  # noqa: E402

ratings = synth.ml25m_like()
dev = torch.device("cuda:0")
if len(sys.argv) > 1:  # python tools/knn_only.py <save_nbrs>: build + truncation
    import time

    from lkpy_amd import _device as D

    ui, iu, _ = _knn_bench.prepare_explicit(ratings)
    dui, diu = D.DeviceCSR.from_scipy(ui, dev), D.DeviceCSR.from_scipy(iu, dev)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = D.iknn_build(dui, diu, 1.0e-6, int(sys.argv[1]))
        torch.cuda.synchronize()
        ts.append(round(time.perf_counter() - t0, 4))
        nnz = int(out.indices.shape[0])
        del out
    print(json.dumps({"save_nbrs": int(sys.argv[1]), "seconds": ts, "nnz": nnz}))
else:
    print(json.dumps(_knn_bench.run(ratings, dev, reps=2)))
