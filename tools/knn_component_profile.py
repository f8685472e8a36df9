#!/usr/bin/env python3
"""Where the host time of `batch.recommend(pipe, 10 000 users, 100)` through the item-kNN components
goes (bench.py's `knn.recommend.through_components` leg): cProfile of one call, cumulative top 25."""
import cProfile
import pstats
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from lkpy_amd import batch as lk_batch  # noqa: E402
from lkpy_amd import synth  # noqa: E402
from lkpy_amd.data import Dataset, Vocabulary  # noqa: E402
from lkpy_amd.pipeline import Pipeline  # noqa: E402

ratings = synth.ml25m_like()
n_u, n_i = ratings.shape
rows = np.repeat(np.arange(n_u, dtype=np.int32), np.diff(ratings.indptr))
ds = Dataset(Vocabulary(np.arange(n_u), "user", reorder=False),
             Vocabulary(np.arange(n_i), "item", reorder=False),
             rows, ratings.indices, {"rating": ratings.data})
pipe = Pipeline.load_config(ROOT / "tests" / "golden" / "pipelines" / "iknn-explicit.toml")
scorer = pipe.node("scorer").component
scorer.config.save_nbrs, scorer.config.max_nbrs, scorer.config.min_nbrs = 100, 100, 1
pipe.train(ds)
users = np.random.default_rng(43).choice(n_u, 10000, replace=False)
lk_batch.recommend(pipe, users[:256], 100)
for _ in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lk_batch.recommend(pipe, users, 100)
    print("call", round((time.perf_counter() - t0) * 1e3, 3), "ms")
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
lk_batch.recommend(pipe, users, 100)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
