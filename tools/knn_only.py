#!/usr/bin/env python3
"Item-kNN build timing only (ML-25M-shaped synthetic): python tools/knn_only.py"
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lkpy_amd import _knn_bench, synth  # noqa: E402

print(json.dumps(_knn_bench.run(synth.ml25m_like(), torch.device("cuda:0"), reps=2)))
