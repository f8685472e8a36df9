"""
Batch inference that routes whole query batches to the fused kernels (SURVEY.md section 8f,
rank 1): the reference's ``batch.recommend`` / ``batch.predict`` loop queries in Python
(src/lenskit/batch/_runner.py:259-345); results here are keyed by user like its
``ItemListCollection``.
"""

from __future__ import annotations

import numpy as np

from .data import ItemList, ItemListCollection, RecQuery
from .pipeline import Pipeline


def recommend(pipe: Pipeline, users, n: int, *, batch_size: int = 16384) -> ItemListCollection:
    """Ordered lists of ``n`` recommendations as an ``ItemListCollection`` keyed by ``user_id``
    (what ``BatchResults.output("recommendations")`` is in the reference,
    src/lenskit/batch/_runner.py:157-191): ``out.lookup(user)`` / ``out.lookup(user_id=user)``,
    iteration over ``(key, list)``, ``out.to_df()``."""
    scorer = pipe.node("scorer").component
    lookup = pipe.node("history-lookup").component
    if hasattr(scorer, "recommend_batch") and hasattr(lookup, "batch") and \
            getattr(scorer, "accepts_history_batch", False):
        # the whole batch by user number: the histories are rows of the HBM-resident training
        # matrix, no per-query host work (an id ARRAY stays an array); the lists are built when
        # somebody looks at them
        ids = users if isinstance(users, np.ndarray) else np.asarray(list(users))
        idx, sc = [], []
        for s in range(0, len(ids), batch_size):
            i, v = scorer.recommend_batch(lookup.batch(ids[s:s + batch_size]), n)
            idx.append(i)
            sc.append(v)
        if not idx:
            return ItemListCollection(("user_id",))
        one = len(idx) == 1
        return ItemListCollection.from_arrays(ids, idx[0] if one else np.concatenate(idx),
                                              sc[0] if one else np.concatenate(sc),
                                              scorer.items, key=("user_id",))
    users = list(users)
    out = {}
    if hasattr(scorer, "recommend_batch"):
        for s in range(0, len(users), batch_size):
            chunk = users[s:s + batch_size]
            queries = [lookup(RecQuery.create(u)) for u in chunk]
            idx, sc = scorer.recommend_batch(queries, n)
            for u, i, v in zip(chunk, idx, sc):
                keep = i >= 0
                out[u] = ItemList(item_nums=i[keep], vocabulary=scorer.items, scores=v[keep],
                                  ordered=True)
    else:
        for u in users:
            out[u] = pipe.run("recommender", query=u, n=n)
    return ItemListCollection.from_dict(out, key=("user_id",))


def predict(pipe: Pipeline, pairs: dict) -> ItemListCollection:
    """Scores for each user's items (``rating-predictor`` semantics) as an ``ItemListCollection``
    keyed by ``user_id`` (``BatchResults.output("predictions")``)."""
    scorer = pipe.node("scorer").component
    lookup = pipe.node("history-lookup").component
    users = list(pairs)
    lists = [pairs[u] if isinstance(pairs[u], ItemList) else ItemList(np.asarray(pairs[u]))
             for u in users]
    if hasattr(scorer, "score_batch"):
        queries = [lookup(RecQuery.create(u)) for u in users]
        scored = scorer.score_batch(queries, lists)
        merger = pipe.nodes.get("rating-merger")
        if merger is not None:
            fb = pipe.node("fallback-predictor").component
            scored = [merger.component(primary=s, backup=fb(q, il))
                      for s, q, il in zip(scored, queries, lists)]
        return ItemListCollection.from_dict(dict(zip(users, scored)), key=("user_id",))
    return ItemListCollection.from_dict(
        {u: pipe.run("rating-predictor", query=u, items=il) for u, il in zip(users, lists)},
        key=("user_id",))
