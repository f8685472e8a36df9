#!/usr/bin/env python3
"""Generate the committed golden fixtures from the read-only reference checkout.

Run ONCE in the build container (``/root/reference`` does not exist on the GPU
box; nothing at test/bench time reads it):

    python tests/golden/make_fixtures.py

Outputs (all under tests/golden/):

* ``ml_small.npz`` -- the ml-latest-small interaction triples the reference's
  own fixtures use (``src/lenskit/testing/_movielens.py:32,46-103`` loads
  ``data/ml-latest-small``), stored as numeric arrays: ``user_id``, ``item_id``,
  ``rating`` (ratings.csv order) and ``all_item_ids`` (every movies.csv id; the
  modern MovieLens loader registers *all* of them as items,
  ``src/lenskit/data/sources/movielens.py:327-345``, so empty item rows exist).
* ``pipelines/als-implicit.toml``, ``iknn-explicit.toml``, ``als-explicit.toml`` -- verbatim copies of the
  reference's pipeline definitions (``pipelines/*.toml``), test INPUTS: the backend must load
  and run them unchanged (tests/test_gpu_pipeline.py, tests/test_host_logic.py).
* ``item-item-preds.csv`` -- the reference's golden item-kNN predictions
  (``tests/models/item-item-preds.csv``, consumed by
  ``tests/models/test_knn_item_item.py:413-453``): 1288 (user,item,prediction)
  rows for ItemKNNScorer(k=20, min_sim=1e-6) on ml-latest-small.
* ``user-user-preds.csv`` -- the reference's golden user-kNN predictions
  (``tests/models/user-user-preds.csv``): 1756 rows for UserKNNScorer(k=30, min_sim=1e-6).
"""
from __future__ import annotations

import shutil
from pathlib import Path

import numpy as np
import pandas as pd

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent


def main():
    ratings = pd.read_csv(REF / "data/ml-latest-small/ratings.csv")
    movies = pd.read_csv(REF / "data/ml-latest-small/movies.csv")
    np.savez_compressed(
        OUT / "ml_small.npz",
        user_id=ratings["userId"].to_numpy(np.int32),
        item_id=ratings["movieId"].to_numpy(np.int32),
        rating=ratings["rating"].to_numpy(np.float32),
        all_item_ids=movies["movieId"].to_numpy(np.int32),
    )
    shutil.copyfile(REF / "tests/models/item-item-preds.csv", OUT / "item-item-preds.csv")
    # the reference's golden user-kNN predictions (tests/models/user-user-preds.csv, consumed by
    # tests/models/test_knn_user_user.py:204-237): UserKNNScorer(k=30, min_sim=1e-6)
    shutil.copyfile(REF / "tests/models/user-user-preds.csv", OUT / "user-user-preds.csv")
    # the two pipeline definitions the north star names: they must load and run UNCHANGED
    (OUT / "pipelines").mkdir(exist_ok=True)
    for name in ("als-implicit.toml", "iknn-explicit.toml", "als-explicit.toml"):
        shutil.copyfile(REF / "pipelines" / name, OUT / "pipelines" / name)
    print("ratings", len(ratings), "items", len(movies))


if __name__ == "__main__":
    main()
