"""
Plumbing components of the standard pipelines -- mirror of ``lenskit.basic`` as far as the
two target TOMLs need it (host NumPy; not on the accelerated path except ``TopNRanker``):

``UserTrainingHistoryLookup``       src/lenskit/basic/history.py:37-95
``TrainingItemsCandidateSelector``  src/lenskit/basic/candidates.py:50-94
``TopNRanker``                      src/lenskit/basic/topn.py:32-69 (-> ItemList.top_n ->
                                    the GPU top-N kernel)
``BiasScorer`` / ``FallbackScorer`` src/lenskit/basic/bias.py, composite.py (std:topn-predict)
"""

from __future__ import annotations

import numpy as np
from pydantic import BaseModel

from .data import Dataset, ItemList, RecQuery, Vocabulary
from .pipeline import Component
from .training import TrainingOptions


class LookupConfig(BaseModel):
    interaction_class: str | None = None


class UserTrainingHistoryLookup(Component):
    config: LookupConfig

    def is_trained(self):
        return hasattr(self, "interactions")

    def train(self, data: Dataset, options: TrainingOptions = TrainingOptions()):
        self.interactions = data.interactions(self.config.interaction_class).matrix()

    def __call__(self, query) -> RecQuery:
        query = RecQuery.create(query)
        if query.user_id is None or self.interactions is None:
            return query
        id_type = self.interactions.row_vocabulary.ids().dtype
        if isinstance(query.user_id, str) and issubclass(id_type.type, np.number):
            query.user_id = id_type.type(query.user_id)
        if query.history_items is None:
            query.history_items = self.interactions.row_items(query.user_id)
        return query


class TrainingItemsCandidateConfig(BaseModel):
    exclude: str | None = "query"


class TrainingItemsCandidateSelector(Component):
    config: TrainingItemsCandidateConfig

    def is_trained(self):
        return hasattr(self, "items_")

    def train(self, data: Dataset, options: TrainingOptions = TrainingOptions()):
        self.items_ = data.items

    def __call__(self, query) -> ItemList:
        query = RecQuery.create(query)
        items = ItemList.from_vocabulary(self.items_)
        exclude = query.query_items if self.config.exclude in ("query", "all") else None
        if exclude is not None and len(exclude) > 0:
            nums = exclude.numbers(vocabulary=self.items_, missing="negative")
            items = items.remove(numbers=nums[nums >= 0])
        return items


class TopNConfig(BaseModel):
    n: int | None = None


class TopNRanker(Component):
    config: TopNConfig

    def __call__(self, *, items: ItemList, n: int | None = None) -> ItemList:
        if n is None:
            n = self.config.n or -1
        return items.top_n(n)


class BiasConfig(BaseModel):
    damping: float | tuple[float, float] | dict[str, float] = 0.0
    entities: list[str] = ["user", "item"]


def _damping(d, which):
    if isinstance(d, dict):
        return float(d.get(which, 0.0))
    if isinstance(d, (tuple, list)):
        return float(d[0] if which == "user" else d[1])
    return float(d)


class BiasScorer(Component):
    """``BiasScorer`` / ``BiasModel.learn`` (src/lenskit/basic/bias.py:85-250):
    score = mu + b_i + b_u."""

    config: BiasConfig

    def is_trained(self):
        return hasattr(self, "global_bias")

    def train(self, data: Dataset, options: TrainingOptions = TrainingOptions()):
        ratings = data.interaction_matrix(format="scipy", layout="coo", field="rating")
        nrows, ncols = ratings.shape
        self.global_bias = float(np.mean(ratings.data))
        centered = ratings.data - self.global_bias
        self.items, self.users = None, None
        self.item_biases, self.user_biases = None, None
        if "item" in self.config.entities:
            counts = np.full(ncols, _damping(self.config.damping, "item"))
            sums = np.zeros(ncols)
            np.add.at(counts, ratings.col, 1)
            np.add.at(sums, ratings.col, centered)
            i_bias = np.zeros(ncols, dtype=np.float32)
            np.divide(sums, counts, out=i_bias, where=counts > 0)
            self.items, self.item_biases = data.items, i_bias
            centered = centered - i_bias[ratings.col]
        if "user" in self.config.entities:
            counts = np.full(nrows, _damping(self.config.damping, "user"))
            sums = np.zeros(nrows)
            np.add.at(counts, ratings.row, 1)
            np.add.at(sums, ratings.row, centered)
            u_bias = np.zeros(nrows, dtype=np.float32)
            np.divide(sums, counts, out=u_bias, where=counts > 0)
            self.users, self.user_biases = data.users, u_bias

    def __call__(self, query, items: ItemList) -> ItemList:
        query = RecQuery.create(query)
        scores = np.full(len(items), self.global_bias, dtype=np.float32)
        if self.item_biases is not None:
            idx = items.numbers(vocabulary=self.items, missing="negative")
            m = idx >= 0
            scores[m] += self.item_biases[idx[m]]
        hist = query.history_items
        ratings = hist.field("rating") if hist is not None else None
        if self.users is not None:
            if ratings is not None:
                uoff = np.asarray(ratings, dtype=np.float64) - self.global_bias
                if self.item_biases is not None:
                    r_idx = hist.numbers(vocabulary=self.items, missing="negative")
                    rm = r_idx >= 0
                    uoff[rm] -= self.item_biases[r_idx[rm]]
                ub = np.sum(uoff) / (np.sum(np.isfinite(uoff)) +
                                     _damping(self.config.damping, "user"))
                scores += 0 if np.isnan(ub) else np.float32(ub)
            elif query.user_id is not None:
                uno = self.users.number(query.user_id, missing=None)
                if uno is not None:
                    scores += self.user_biases[uno]
        return ItemList(items, scores=scores)


class FallbackScorer(Component):
    "Primary scores, back-filled from the backup where missing (basic/composite.py)."

    def __call__(self, primary: ItemList, backup: ItemList) -> ItemList:
        ps = np.array(primary.scores(), dtype=np.float32, copy=True)
        missing = np.isnan(ps)
        if np.any(missing):
            b = backup.scores()
            if np.array_equal(primary.ids(), backup.ids()):
                ps[missing] = b[missing]
            else:
                lookup = dict(zip(backup.ids().tolist(), b.tolist()))
                for i in np.flatnonzero(missing):
                    ps[i] = lookup.get(primary.ids()[i], np.nan)
        return ItemList(primary, scores=ps, is_fallback=missing)
