#!/usr/bin/env python3
"cProfile of ImplicitMFScorer.train on the ML-25M-shaped synthetic (after a warm-up fit): python tools/fit_profile.py"
import cProfile
import pstats
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lkpy_amd import synth  # noqa: E402
from lkpy_amd.als import ImplicitMFScorer  # noqa: E402
from lkpy_amd.data import Dataset, Vocabulary  # noqa: E402
from lkpy_amd.training import TrainingOptions  # noqa: E402

ratings = synth.ml25m_like()
n_users, n_items = ratings.shape
rows = np.repeat(np.arange(n_users, dtype=np.int32), np.diff(ratings.indptr))
ds = Dataset(Vocabulary(np.arange(n_users), "user", reorder=False),
             Vocabulary(np.arange(n_items), "item", reorder=False),
             rows, ratings.indices, {"rating": ratings.data})
for rep in range(2):
    scorer = ImplicitMFScorer(embedding_size=64, epochs=20, weight=40.0)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    scorer.train(ds, TrainingOptions(rng=42))
    torch.cuda.synchronize()
    pr.disable()
    print("fit seconds", round(time.perf_counter() - t0, 4), flush=True)
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
