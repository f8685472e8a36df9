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


    # -- whole batches of queries --------------------------------
    def __getstate__(self):
        st = dict(self.__dict__)
        st.pop("_dev", None)
        return st

    def _device_matrix(self):
        "The training matrix resident in HBM (uploaded once, never pickled): offsets, items, ratings."
        st = self.__dict__.get("_dev")
        if st is None:
            from . import _device as D

            ds = self.interactions._ds
            dev = D.device()
            rat = ds._attrs.get("rating")
            st = {"device": dev,
                  "csr": D.DeviceCSR.from_arrays(
                      ds._indptr, ds._cols,
                      np.zeros(0, np.float32) if rat is None else rat,
                      (ds.user_count, ds.item_count), dev),
                  "has_ratings": rat is not None}
            if rat is None:
                st["csr"].values = None
            self._dev = st
        return st

    def batch(self, user_ids) -> "HistoryBatch":
        """
        The training histories of MANY users at once: what ``__call__`` does per query
        (src/lenskit/basic/history.py:77-95: attach ``interactions.row_items(user_id)`` as the
        query's history) for a whole batch, without building an ``ItemList`` per user -- the user
        numbers come from one vectorised vocabulary lookup and the rows are cut out of the
        HBM-resident training matrix by one kernel when a scorer asks for them
        (:meth:`HistoryBatch.csr`).  Unknown users get empty histories, like the reference.
        """
        ids = np.asarray(list(user_ids) if not isinstance(user_ids, np.ndarray) else user_ids)
        vocab = self.interactions.row_vocabulary
        id_type = vocab.ids().dtype
        if ids.dtype.kind in "US" and issubclass(id_type.type, np.number):
            ids = ids.astype(id_type)  # (history.py:85-87: string ids for a numeric vocabulary)
        nums = vocab.numbers(ids, missing="negative") if len(ids) else np.zeros(0, np.int32)
        return HistoryBatch(self, ids, nums)


class HistoryBatch:
    """
    The histories of a batch of queries as rows of the training matrix (user numbers), not as
    one ``ItemList`` per user: ``len``, ``user_ids``, ``user_nums`` (-1 = unknown user),
    ``lengths`` (host), ``items`` (the vocabulary the item numbers refer to), and :meth:`csr` --
    the histories' CSR in HBM, cut out by ``lk_csr_gather_rows``.  ``queries()`` gives the plain
    per-query objects for scorers without a batched path.
    """

    def __init__(self, lookup: "UserTrainingHistoryLookup", user_ids: np.ndarray,
                 user_nums: np.ndarray):
        self.lookup = lookup
        self.user_ids = user_ids
        self.user_nums = np.ascontiguousarray(user_nums, dtype=np.int32)
        ds = lookup.interactions._ds
        self.items: Vocabulary = ds.items
        self.users: Vocabulary = ds.users
        hp = ds._indptr
        safe = np.maximum(self.user_nums, 0)
        self.lengths = np.where(self.user_nums >= 0, hp[safe + 1] - hp[safe], 0).astype(np.int64)
        self._cache: dict = {}

    def __len__(self):
        return len(self.user_nums)

    @property
    def has_ratings(self) -> bool:
        return "rating" in self.lookup.interactions._ds._attrs

    def csr(self, *, use_ratings: bool = False, scale: float = 1.0, with_values: bool = True,
            col_bias=None):
        """Device CSR (queries x items, int64 offsets; ``h_indptr`` = the host copy) of the
        histories; values = ``rating * scale`` (``use_ratings``) or the constant ``scale``, with
        ``col_bias`` (device f32 per item) subtracted from the rating first."""
        from . import _device as D

        key = (bool(use_ratings), float(scale), bool(with_values),
               None if col_bias is None else int(col_bias.data_ptr()))
        hit = self._cache.get(key)
        if hit is not None:
            return hit
        st = self.lookup._device_matrix()
        src = st["csr"]
        if use_ratings and not st["has_ratings"]:
            raise ValueError("no ratings in user items")  # (_implicit.py:85-86)
        if not use_ratings and src.values is not None:
            src = D.DeviceCSR(src.indptr, src.indices, None, src.shape, src.h_indptr)
        out = D.gather_rows(src, self.user_nums, scale=scale, with_values=with_values,
                            col_bias=col_bias)
        self._cache[key] = out
        return out

    def subset(self, mask: np.ndarray) -> "HistoryBatch":
        return HistoryBatch(self.lookup, self.user_ids[mask], self.user_nums[mask])

    def queries(self) -> list:
        return [self.lookup(RecQuery.create(u.item() if isinstance(u, np.generic) else u))
                for u in self.user_ids]


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


class BiasModel:
    """
    User-item bias model  b_ui = b_g + b_i + b_u  with Bayesian damping
    (``BiasModel``, src/lenskit/basic/bias.py:35-275): reusable by components that normalise
    ratings (the biased-MF ALS) as well as by :class:`BiasScorer`.
    """

    def __init__(self, damping=0.0, global_bias: float = 0.0):
        self.damping = damping
        self.global_bias = float(global_bias)
        self.items = self.users = None
        self.item_biases = self.user_biases = None

    @classmethod
    def learn(cls, data: Dataset, damping=0.0, *, entities=("user", "item")) -> "BiasModel":
        "bias.py:83-150, call for call (float64 sums, float32 biases)."
        ratings = data.interaction_matrix(format="scipy", layout="coo", field="rating")
        nrows, ncols = ratings.shape
        model = cls(damping, float(np.mean(ratings.data)))
        centered = ratings.data - model.global_bias
        if "item" in entities:
            counts = np.full(ncols, _damping(damping, "item"))
            sums = np.zeros(ncols)
            np.add.at(counts, ratings.col, 1)
            np.add.at(sums, ratings.col, centered)
            i_bias = np.zeros(ncols, dtype=np.float32)
            np.divide(sums, counts, out=i_bias, where=counts > 0)
            model.items, model.item_biases = data.items, i_bias
            centered = centered - i_bias[ratings.col]
        if "user" in entities:
            counts = np.full(nrows, _damping(damping, "user"))
            sums = np.zeros(nrows)
            np.add.at(counts, ratings.row, 1)
            np.add.at(sums, ratings.row, centered)
            u_bias = np.zeros(nrows, dtype=np.float32)
            np.divide(sums, counts, out=u_bias, where=counts > 0)
            model.users, model.user_biases = data.users, u_bias
        return model

    def compute_for_items(self, items: ItemList, user_id=None, user_items: ItemList | None = None,
                          *, bias: float | None = None):
        """
        bias.py:166-241: composite biases of ``items``; with ``bias`` (a known user bias) only
        the scores are returned, otherwise (scores, user_bias) with the user bias taken from
        ``user_items``' ratings when present, else from the stored value of ``user_id``.
        """
        scores = np.full(len(items), self.global_bias, dtype=np.float32)
        if self.item_biases is not None:
            idx = items.numbers(vocabulary=self.items, missing="negative")
            m = idx >= 0
            scores[m] += self.item_biases[idx[m]]
        if bias is not None:
            return scores + bias
        ratings = user_items.field("rating") if user_items is not None else None
        user_bias = 0.0
        if self.users is not None:
            if ratings is not None:
                uoff = np.asarray(ratings, dtype=np.float64) - self.global_bias
                if self.item_biases is not None:
                    r_idx = user_items.numbers(vocabulary=self.items, missing="negative")
                    rm = r_idx >= 0
                    uoff[rm] -= self.item_biases[r_idx[rm]]
                user_bias = np.sum(uoff) / (np.sum(np.isfinite(uoff)) +
                                            _damping(self.damping, "user"))
                if np.isnan(user_bias):
                    user_bias = 0
                scores += np.float32(user_bias)
            elif user_id is not None:
                uno = self.users.number(user_id, missing=None)
                if uno is not None:
                    user_bias = self.user_biases[uno]
                    scores += user_bias
        return scores, user_bias

    def transform_matrix(self, matrix):
        "bias.py:246-275: subtract the biases from a COO ratings matrix."
        import scipy.sparse as sps

        values = matrix.data - self.global_bias
        if self.item_biases is not None:
            values -= self.item_biases[matrix.col]
        if self.user_biases is not None:
            values -= self.user_biases[matrix.row]
        return sps.coo_array((values, (matrix.row, matrix.col)), shape=matrix.shape)


class BiasScorer(Component):
    """``BiasScorer`` (src/lenskit/basic/bias.py:278-360) over :class:`BiasModel`:
    score = mu + b_i + b_u."""

    config: BiasConfig

    def is_trained(self):
        return hasattr(self, "model")

    def train(self, data: Dataset, options: TrainingOptions = TrainingOptions()):
        self.model = BiasModel.learn(data, self.config.damping, entities=self.config.entities)

    # the learned parameters, under the names the reference's scorer exposes
    global_bias = property(lambda self: self.model.global_bias)
    item_biases = property(lambda self: self.model.item_biases)
    user_biases = property(lambda self: self.model.user_biases)
    items = property(lambda self: self.model.items)
    users = property(lambda self: self.model.users)

    def __call__(self, query, items: ItemList) -> ItemList:
        query = RecQuery.create(query)
        scores, _ = self.model.compute_for_items(items, query.user_id, query.history_items)
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
