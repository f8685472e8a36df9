#!/usr/bin/env python3
"""Item-kNN recommend leg only (ML-25M-shaped synthetic, save_nbrs = 100 model, top-100 for
10 000 users): python tools/knn_recommend_only.py   -- what tools/prof_knnrec.sh profiles."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lkpy_amd import _device as D  # noqa: E402
from lkpy_amd import _knn_bench, synth  # noqa: E402

ratings = synth.ml25m_like()
dev = torch.device("cuda:0")
dui, diu, means, _ = D.iknn_prepare(ratings, True, dev)
sims = D.iknn_build(dui, diu, 1.0e-6, 100)
del dui, diu
res = _knn_bench._recommend_leg(D, ratings, means, sims, dev)
print(json.dumps(res))
