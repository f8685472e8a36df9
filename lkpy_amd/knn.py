"""
Component seam for neighbourhood models -- item-based k-NN and user-based
k-NN (mirror of ``lenskit.knn.UserKNNScorer``, src/lenskit/knn/user.py:25-316) and EASE (end of
the file, mirror of ``lenskit.knn.EASEScorer``, src/lenskit/knn/ease.py).

Item-based k-NN: mirror of ``lenskit.knn.ItemKNNScorer`` /
``ItemKNNConfig`` (src/lenskit/knn/item.py:41-295).  Matrix preparation is the reference's
own SciPy code path (item-mean centring, L2 normalisation); the similarity build and the
scoring run in the HIP kernels.
"""

from __future__ import annotations

import warnings
from typing import Literal

import numpy as np
import scipy.sparse as sps
import torch
from pydantic import AliasChoices, BaseModel, Field, PositiveFloat, PositiveInt, field_validator

from . import _device as D
from .data import Dataset, ItemList, RecQuery, SparseRowArray, Vocabulary
from .pipeline import Component
from .training import TrainingOptions


class DataWarning(UserWarning):
    "``lenskit.diagnostics.DataWarning``"


class ItemKNNConfig(BaseModel, extra="forbid"):
    "src/lenskit/knn/item.py:41-84"

    max_nbrs: PositiveInt = Field(20, validation_alias=AliasChoices("max_nbrs", "nnbrs", "k"))
    min_nbrs: PositiveInt = 1
    min_sim: PositiveFloat = 1.0e-6
    save_nbrs: PositiveInt | None = None
    feedback: Literal["explicit", "implicit"] = "explicit"
    block_size: int = 250

    @field_validator("min_sim", mode="after")
    @staticmethod
    def clamp_min_sim(sim) -> float:
        return max(sim, float(np.finfo(np.float64).smallest_normal))

    @property
    def explicit(self) -> bool:
        return self.feedback == "explicit"


class ItemKNNScorer(Component):
    config: ItemKNNConfig

    items: Vocabulary
    item_means: np.ndarray | None
    item_counts: np.ndarray
    sim_matrix: SparseRowArray

    def is_trained(self):
        return hasattr(self, "sim_matrix")

    def __getstate__(self):
        st = dict(self.__dict__)
        st.pop("_dev", None)
        return st

    def train(self, data: Dataset, options: TrainingOptions = TrainingOptions()):
        field = "rating" if self.config.explicit else None
        rmat = data.interactions().matrix().scipy(field, layout="coo").astype(np.float32)
        n_rows, n_items = rmat.shape
        dev = D.device()
        # centring + normalisation (item.py:142-156,202-228) on the device, bit-identical to
        # the SciPy calls of the reference (see _device.iknn_prepare)
        dui, diu, means, all_zero = D.iknn_prepare(rmat, self.config.explicit, dev)
        if all_zero:
            warnings.warn("Ratings seem to have the same value, centering is not recommended.",
                          DataWarning)
        out = D.iknn_build(dui, diu, self.config.min_sim, self.config.save_nbrs)
        self.items = data.items
        self.item_means = None if means is None else np.asarray(means)
        offsets = out.indptr.cpu().numpy()
        self.item_counts = np.diff(offsets)
        # Arrow extension array, int64 offsets (item.py:176-177: LargeList -> from_array); an
        # unbounded ML-25M model is 9.2 GB: D.to_host moves it at PCIe speed (lk_download)
        self.sim_matrix = SparseRowArray.from_arrays(
            offsets, D.to_host(out.indices, index_bound=out.shape[1]), D.to_host(out.values),
            shape=(n_items, n_items))
        import pyarrow as pa

        assert pa.types.is_large_list(self.sim_matrix.type.storage_type)
        self._dev = {"sims": out, "device": dev}

    def _device_sims(self):
        dev = getattr(self, "_dev", None)
        if dev is None:
            d = D.device()
            from .matrix import csr_arrays

            so, si, sv, shape = csr_arrays(self.sim_matrix)
            dev = {"device": d,
                   "sims": D.DeviceCSR(
                       torch.from_numpy(np.array(so, dtype=np.int64)).to(d),
                       torch.from_numpy(np.array(si, dtype=np.int32)).to(d),
                       torch.from_numpy(np.array(sv, dtype=np.float32)).to(d),
                       shape, None)}
            self._dev = dev
        return dev

    def score_batch(self, queries, item_lists) -> list[ItemList]:
        "Score many (query, items) pairs in one kernel launch (item.py:231-295 per pair)."
        st = self._device_sims()
        d = st["device"]
        queries = [RecQuery.create(q) for q in queries]
        r_idx, r_val, r_ptr, t_idx, t_ptr = [], [], [0], [], [0]
        nohist = []
        for q, items in zip(queries, item_lists):
            ratings = q.query_items
            if ratings is None or len(ratings) == 0:
                nohist.append(True)
                r_ptr.append(r_ptr[-1])
            else:
                nohist.append(False)
                ri = ratings.numbers(vocabulary=self.items, missing="negative")
                if self.config.explicit:
                    rv = ratings.field("rating")
                    if rv is None:
                        raise RuntimeError("explicit-feedback scorer must have ratings")
                    rv = np.asarray(rv).astype(np.float32, copy=True)
                    m = ri >= 0
                    rv[m] -= self.item_means[ri[m]]  # mean-centre (item.py:268-271)
                    r_val.append(rv)
                r_idx.append(ri)
                r_ptr.append(r_ptr[-1] + len(ri))
            ti = items.numbers(vocabulary=self.items, missing="negative")
            t_idx.append(ti)
            t_ptr.append(t_ptr[-1] + len(ti))
        cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)  # noqa: E731
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(d)  # noqa: E731
        rr = to(cat(r_val, np.float32)) if self.config.explicit else None
        s, c = D.iknn_score_batch(st["sims"], to(np.asarray(r_ptr, np.int64)),
                                  to(cat(r_idx, np.int32)), rr, to(np.asarray(t_ptr, np.int64)),
                                  to(cat(t_idx, np.int32)), self.config.max_nbrs,
                                  self.config.min_nbrs)
        s, c = s.cpu().numpy(), c.cpu().numpy()
        out = []
        for qi, items in enumerate(item_lists):
            if nohist[qi]:
                out.append(ItemList(items, scores=np.nan))  # item.py:238-245
                continue
            sc = s[t_ptr[qi]:t_ptr[qi + 1]].copy()
            ti = t_idx[qi]
            if self.config.explicit:
                m = ti >= 0
                sc[m] += self.item_means[ti[m]]  # item.py:282
            out.append(ItemList(items, scores=sc, nbr_counts=c[t_ptr[qi]:t_ptr[qi + 1]]))
        return out

    def __call__(self, query, items: ItemList) -> ItemList:
        return self.score_batch([query], [items])[0]

    def recommend_batch(self, queries, n: int, *, exclude_history: bool = True):
        """
        Top-``n`` lists for many queries at once -- what the ``recommender`` pipeline computes one
        query at a time (src/lenskit/batch/_runner.py:283-308): candidates = every training item
        minus the query's own (src/lenskit/basic/candidates.py:77-94), this scorer over them
        (item.py:231-295, means added back: 282), ``TopNRanker`` (basic/topn.py:45-69).  One
        ``lk_iknn_recommend`` call: scores bit-identical to the reference accumulator's, items
        with fewer than ``min_nbrs`` neighbours never listed, queries without history get empty
        lists (item.py:238-245: all-NaN scores).  Returns (item numbers [B x n] with -1 padding,
        scores [B x n] with NaN padding), like ``ImplicitMFScorer.recommend_batch``.
        """
        from .basic import HistoryBatch

        if isinstance(queries, HistoryBatch) and not (
                queries.items is self.items or queries.items == self.items):
            queries = queries.queries()  # (another item vocabulary: the per-query mapping)
        if isinstance(queries, HistoryBatch):
            return self._recommend_history_batch(queries, n, exclude_history)
        st = self._device_sims()
        d = st["device"]
        queries = [RecQuery.create(q) for q in queries]
        r_idx, r_val, r_ptr = [], [], [0]
        for q in queries:
            ratings = q.query_items
            if ratings is None or len(ratings) == 0:
                r_ptr.append(r_ptr[-1])
                continue
            ri = ratings.numbers(vocabulary=self.items, missing="negative")
            if self.config.explicit:
                rv = ratings.field("rating")
                if rv is None:
                    raise RuntimeError("explicit-feedback scorer must have ratings")
                rv = np.asarray(rv).astype(np.float32, copy=True)
                m = ri >= 0
                rv[m] -= self.item_means[ri[m]]  # mean-centre (item.py:268-271)
                r_val.append(rv)
            r_idx.append(ri)
            r_ptr.append(r_ptr[-1] + len(ri))
        cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)  # noqa: E731
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(d)  # noqa: E731
        idx = cat(r_idx, np.int32)
        ptr = np.asarray(r_ptr, np.int64)
        # hits per query: the similarity-row lengths of its history items, summed
        counts = np.asarray(self.item_counts, dtype=np.int64)
        per = np.where(idx >= 0, counts[np.maximum(idx, 0)], 0)
        csum = np.concatenate([[0], np.cumsum(per)])
        hits = csum[ptr[1:]] - csum[ptr[:-1]]
        # heaviest queries first (the launch is as long as its longest task chain); undone below
        order = np.argsort(-hits, kind="stable")
        lens = np.diff(ptr)[order]
        optr = np.zeros(len(order) + 1, np.int64)
        np.cumsum(lens, out=optr[1:])
        take = np.concatenate([np.arange(ptr[q], ptr[q + 1]) for q in order]) if len(idx) else \
            np.zeros(0, np.int64)
        bias = st.get("means")
        if bias is None and self.config.explicit and self.item_means is not None:
            bias = st["means"] = to(np.asarray(self.item_means, dtype=np.float32))
        rr = to(cat(r_val, np.float32)[take]) if self.config.explicit else None
        oi, osc = D.iknn_recommend(st["sims"], to(optr), to(idx[take]), rr,
                                   bias if self.config.explicit else None, self.config.max_nbrs,
                                   self.config.min_nbrs, n, hits[order], exclude_history)
        inv = np.empty_like(order)
        inv[order] = np.arange(len(order))
        return oi.cpu().numpy()[inv], osc.cpu().numpy()[inv]


    accepts_history_batch = True  # recommend_batch takes a lkpy_amd.basic.HistoryBatch

    def _recommend_history_batch(self, batch, n: int, exclude_history: bool):
        """
        ``recommend_batch`` for training histories by user number (``UserTrainingHistoryLookup.
        batch``): nothing is done per query on the host.  The histories -- in the training rows'
        order, which is what the reference's lookup hands the scorer (basic/history.py:77-95) --
        are cut out of the HBM-resident training matrix with the ratings mean-centred on the way
        (``lk_csr_gather_rows`` with the item means as column bias: item.py:268-271); a query's
        hit count (the summed similarity-row lengths of its history items, what the launch is
        balanced by) is a property of the USER, computed once for every training user on the
        device; the batch goes heaviest query first and the lists come back in the caller's order.
        """
        st = self._device_sims()
        d = st["device"]
        if self.config.explicit and not batch.has_ratings:
            raise RuntimeError("explicit-feedback scorer must have ratings")
        bias = st.get("means")
        if bias is None and self.config.explicit and self.item_means is not None:
            bias = st["means"] = torch.from_numpy(
                np.asarray(self.item_means, dtype=np.float32)).to(d)
        key = ("user_hits", id(batch.lookup))
        user_hits = st.get(key)
        if user_hits is None:
            # hits of every training user: counts[item] summed over the user's row, on the device
            mat = batch.lookup._device_matrix()["csr"]
            cnt = torch.from_numpy(np.asarray(self.item_counts, dtype=np.int64)).to(d)
            csum = torch.zeros(mat.indices.numel() + 1, dtype=torch.int64, device=d)
            torch.cumsum(cnt[mat.indices.long()], 0, out=csum[1:])
            ptr = mat.indptr.long()
            user_hits = st[key] = (csum[ptr[1:]] - csum[ptr[:-1]]).cpu().numpy()
        nums = batch.user_nums
        hits = np.where(nums >= 0, user_hits[np.maximum(nums, 0)], 0).astype(np.int64)
        order = np.argsort(-hits, kind="stable")  # heaviest queries first; undone below
        sub = batch.subset(order)
        hist = sub.csr(use_ratings=self.config.explicit, scale=1.0,
                       col_bias=bias if self.config.explicit else None,
                       with_values=self.config.explicit)
        oi, osc = D.iknn_recommend(st["sims"], hist.indptr, hist.indices,
                                   hist.values if self.config.explicit else None,
                                   bias if self.config.explicit else None, self.config.max_nbrs,
                                   self.config.min_nbrs, n, hits[order], exclude_history)
        inv_h = np.empty_like(order)
        inv_h[order] = np.arange(len(order))  # (the inverse permutation: O(n), no second sort)
        inv = torch.from_numpy(inv_h).to(d)
        both = torch.cat([oi.view(torch.float32), osc], dim=1)[inv]
        host = D.to_host(both)
        cols = oi.shape[1]
        return host[:, :cols].view(np.int32), host[:, cols:]


# ---------------------------------------------------------------------------------------
# User-based k-NN
# ---------------------------------------------------------------------------------------


class UserKNNConfig(BaseModel, extra="forbid"):
    "src/lenskit/knn/user.py:39-71"

    max_nbrs: PositiveInt = Field(20, validation_alias=AliasChoices("max_nbrs", "nnbrs", "k"))
    min_nbrs: PositiveInt = 1
    min_sim: PositiveFloat = 1.0e-6
    feedback: Literal["explicit", "implicit"] = "explicit"

    @field_validator("min_sim", mode="after")
    @staticmethod
    def clamp_min_sim(sim) -> float:
        return max(sim, float(np.finfo(np.float64).smallest_normal))

    @property
    def explicit(self) -> bool:
        return self.feedback == "explicit"


class UserKNNScorer(Component):
    """
    User-user nearest-neighbour collaborative filtering (``UserKNNScorer``,
    src/lenskit/knn/user.py:74-316).  "Training" memorises the mean-centred and the
    row-normalised rating matrices with the reference's own SciPy calls; scoring runs on the
    device for whole batches of queries: neighbour similarities = normalised matrix x query
    vector (``lk_csr_rows_dot``), neighbours with sim >= min_sim in ascending user order,
    per-item accumulation of the ``max_nbrs`` most similar raters (``lk_uknn_score_batch``).
    """

    config: UserKNNConfig

    users: Vocabulary
    items: Vocabulary
    user_means: np.ndarray | None
    user_vectors: sps.csr_array
    user_ratings: SparseRowArray

    def is_trained(self):
        return hasattr(self, "user_ratings")

    def __getstate__(self):
        st = dict(self.__dict__)
        st.pop("_dev", None)
        return st

    def train(self, data: Dataset, options: TrainingOptions = TrainingOptions()):
        "user.py:122-168: centre by user mean (explicit), normalise rows, keep both matrices"
        import scipy.sparse.linalg as spla

        rmat = data.interactions().matrix().scipy(
            attribute="rating" if self.config.explicit else None).astype(np.float32)
        means = None
        if self.config.explicit:
            counts = np.diff(rmat.indptr)
            sums = rmat.sum(axis=1)
            means = np.zeros(sums.shape, dtype=np.float32)
            np.divide(sums, counts, out=means, where=counts > 0)
            rmat.data = rmat.data - np.repeat(means, counts)
            if np.allclose(rmat.data, 0.0):
                warnings.warn("Ratings seem to have the same value, centering is not "
                              "recommended.", DataWarning)
        norms = spla.norm(rmat, 2, axis=1)
        cmat = rmat / np.maximum(norms, np.finfo("f4").smallest_normal).reshape(-1, 1)
        self.user_vectors = sps.csr_array(cmat.tocsr())
        self.user_ratings = SparseRowArray.from_scipy(rmat, values=self.config.explicit)
        self.users = data.users
        self.user_means = means
        self.items = data.items
        self.__dict__.pop("_dev", None)

    def _device_state(self):
        st = getattr(self, "_dev", None)
        if st is None:
            from .matrix import csr_arrays

            d = D.device()
            uv = self.user_vectors
            uv.sort_indices()
            ro, ri, rv, shape = csr_arrays(self.user_ratings)
            st = {
                "device": d,
                "vectors": D.DeviceCSR.from_arrays(uv.indptr, uv.indices, uv.data, uv.shape, d),
                "ratings": D.DeviceCSR(
                    torch.from_numpy(np.array(ro, dtype=np.int64)).to(d),
                    torch.from_numpy(np.array(ri, dtype=np.int32)).to(d),
                    None if rv is None else torch.from_numpy(np.array(rv, np.float32)).to(d),
                    shape, None),
            }
            self._dev = st
        return st

    def _user_data(self, query: RecQuery):
        "``_get_user_data`` (user.py:264-307): (user number | None, dense item vector, mean)"
        index = self.users.number(query.user_id, missing=None) \
            if query.user_id is not None else None
        hist = query.query_items
        n_items = len(self.items)
        if hist is None:
            if index is None:
                return None
            uv = self.user_vectors
            row = np.zeros(n_items, dtype=np.float32)
            s, e = uv.indptr[index], uv.indptr[index + 1]
            row[uv.indices[s:e]] = uv.data[s:e]
            umean = float(self.user_means[index]) if self.config.explicit else 0.0
            return index, row, umean
        if len(hist) == 0:
            return None
        ratings = np.zeros(n_items, dtype=np.float32)
        nos = hist.numbers(missing="negative", vocabulary=self.items)
        ok = nos >= 0
        if self.config.explicit:
            urv = hist.field("rating")
            if urv is None:
                return None
            urv = np.require(urv, dtype=np.float32)
            umean = float(urv.mean())
            ratings[nos[ok]] = urv[ok] - umean
        else:
            umean = 0.0
            ratings[nos[ok]] = 1.0
        return index, ratings, umean

    def score_batch(self, queries, item_lists) -> list[ItemList]:
        "Many (query, items) pairs through two kernel launches (user.py:171-262 per pair)."
        st = self._device_state()
        d = st["device"]
        queries = [RecQuery.create(q) for q in queries]
        data = [self._user_data(q) if len(il) > 0 else None
                for q, il in zip(queries, item_lists)]
        live = [i for i, u in enumerate(data) if u is not None]
        out: list[ItemList | None] = [None] * len(queries)
        for i, il in enumerate(item_lists):
            if data[i] is None:
                out[i] = ItemList(il, scores=np.nan)
        if not live:
            return out  # type: ignore[return-value]
        # neighbour similarities for the whole batch: [B x users]
        X = np.stack([data[i][1] for i in live], axis=1)  # [items x B]
        sims = D.csr_rows_dot(st["vectors"], torch.from_numpy(np.ascontiguousarray(X)).to(d))
        for b, i in enumerate(live):
            if data[i][0] is not None:
                sims[b, data[i][0]] = 0.0  # zero out the self-similarity (user.py:199-201)
        mask = sims >= float(np.float32(self.config.min_sim))  # user.py:206 (f32 comparison)
        counts = mask.sum(dim=1)
        nbr_ptr = torch.zeros(len(live) + 1, dtype=torch.int64, device=d)
        nbr_ptr[1:] = torch.cumsum(counts, 0)
        nbr_rows = mask.nonzero()[:, 1].to(torch.int32)  # per query, ascending user number
        nbr_sims = sims[mask]
        t_idx, t_ptr = [], [0]
        for i in live:
            ti = item_lists[i].numbers(vocabulary=self.items, missing="negative")
            t_idx.append(ti)
            t_ptr.append(t_ptr[-1] + len(ti))
        tgt = torch.from_numpy(np.concatenate(t_idx).astype(np.int32)).to(d)
        s, _c = D.uknn_score_batch(st["ratings"], nbr_ptr, nbr_rows.contiguous(),
                                   nbr_sims.contiguous(),
                                   torch.from_numpy(np.asarray(t_ptr, np.int64)).to(d), tgt,
                                   self.config.max_nbrs, self.config.min_nbrs)
        s = s.cpu().numpy()
        has_nbrs = counts.cpu().numpy() > 0
        for b, i in enumerate(live):
            sc = s[t_ptr[b]:t_ptr[b + 1]].copy()
            if not has_nbrs[b]:
                sc[:] = np.nan  # no candidate neighbours (user.py:217-219)
            out[i] = ItemList(item_lists[i], scores=sc + np.float32(data[i][2]))
        return out  # type: ignore[return-value]

    def __call__(self, query, items: ItemList) -> ItemList:
        return self.score_batch([query], [items])[0]


class EASEConfig(BaseModel, extra="forbid"):
    "``EASEConfig`` (src/lenskit/knn/ease.py:36-44)."

    regularization: PositiveFloat = 1
    "Regularization term for EASE."


class EASEScorer(Component):
    """
    Embarrassingly shallow autoencoder (``EASEScorer``, src/lenskit/knn/ease.py:47-170;
    SURVEY.md section 8f rank 4).  Training: the dense item-item co-occurrence Gramian
    ``X^T X + reg I`` is built on the device (the similarity-build kernel on unit values +
    ``lk_ease_gram``), inverted there with the very PyTorch calls of the reference's
    ``_chol_invert_torch`` (``torch.linalg.cholesky_ex`` + ``torch.cholesky_inverse``,
    ease.py:190-208 -- a library factorisation, as in the reference), and turned into the
    weight matrix ``B = inv / -diag(inv)`` with a zero diagonal.  Scoring = sum of the history
    items' weight rows (``lk_ease_score_batch``), whole batches of queries at a time.
    """

    config: EASEConfig

    items: Vocabulary
    weights: np.ndarray

    def is_trained(self):
        return hasattr(self, "weights")

    def __getstate__(self):
        st = dict(self.__dict__)
        st.pop("_dev", None)
        return st

    def train(self, data: Dataset, options: TrainingOptions = TrainingOptions()):
        solver = options.env_var("LK_EASE_SOLVER", None)
        if solver and solver not in ("torch", "scipy"):
            raise ValueError(f"unsupported option: LK_EASE_SOLVER={solver}")  # ease.py:96-97
        if solver == "scipy":
            raise ValueError("LK_EASE_SOLVER=scipy: the device backend has no host solver")
        n_items = data.item_count
        ui = data.interactions().matrix().scipy(attribute=None).astype(np.float32)
        ui = sps.csr_array(ui)
        ui.sum_duplicates()
        ui.data[:] = 1.0  # co-occurrences count (user, item) pairs once
        ui.sort_indices()
        iu = sps.csr_array(ui.T)
        iu.sort_indices()
        d = D.device()
        cooc = D.iknn_build(D.DeviceCSR.from_scipy(ui, d), D.DeviceCSR.from_scipy(iu, d), 0.5)
        counts = torch.from_numpy(np.diff(iu.indptr).astype(np.int32)).to(d)
        gram = D.ease_gram(cooc, counts, float(self.config.regularization))
        del cooc
        with torch.inference_mode():
            decomp, info = torch.linalg.cholesky_ex(gram)
            if info.item():
                # ease.py:202-203
                raise RuntimeError(f"matrix minor {info.item()} is not positive-definite.")
            inv = torch.cholesky_inverse(decomp, out=gram)
            del decomp
            # divide cells by the column's diagonal entry, zero the diagonal (ease.py:140-142)
            inv /= -torch.diagonal(inv).reshape(1, -1).clone()
            inv.fill_diagonal_(0.0)
            mat = inv.cpu().numpy()
        self.items = data.items
        self.weights = mat
        self.__dict__.pop("_dev", None)
        assert self.weights.shape == (n_items, n_items)

    def _device_weights(self):
        w = getattr(self, "_dev", None)
        if w is None:
            w = torch.from_numpy(np.ascontiguousarray(self.weights, dtype=np.float32)).to(
                D.device())
            self._dev = w
        return w

    def score_batch(self, queries, item_lists) -> list[ItemList]:
        "Scores for a batch of (query, items) pairs; one device call for all of them."
        w = self._device_weights()
        hists, ok = [], []
        for query in queries:
            query = RecQuery.create(query)
            q_items = query.query_items
            good = np.empty(0, np.int32)
            if q_items is not None:
                q_inos = q_items.numbers(vocabulary=self.items, missing="negative")
                # a repeated history item counts ONCE: the reference sets q_vec[q_good] = 1.0
                # (src/lenskit/knn/ease.py), it does not add per occurrence
                good = np.unique(q_inos[q_inos >= 0]).astype(np.int32)
            hists.append(good)
            ok.append(len(good) > 0)  # ease.py:150-158: no usable history => all NaN
        ptr = np.zeros(len(hists) + 1, dtype=np.int64)
        np.cumsum([len(h) for h in hists], out=ptr[1:])
        cat = np.concatenate(hists) if hists else np.empty(0, np.int32)
        scores = D.ease_score_batch(torch.from_numpy(ptr).to(w.device),
                                    torch.from_numpy(np.ascontiguousarray(cat)).to(w.device),
                                    w).cpu().numpy()
        out = []
        for i, items in enumerate(item_lists):
            if not ok[i]:
                out.append(ItemList(items, scores=np.nan))
                continue
            t_inos = items.numbers(vocabulary=self.items, missing="negative")
            sc = np.full(len(items), np.nan, dtype=np.float32)
            t_ok = t_inos >= 0
            sc[t_ok] = scores[i][t_inos[t_ok]]
            out.append(ItemList(items, scores=sc))
        return out

    def __call__(self, query, items: ItemList) -> ItemList:
        return self.score_batch([query], [items])[0]
