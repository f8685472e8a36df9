"""
Component seam for ALS matrix factorisation (implicit feedback; the explicit / biased-MF model
is at the end of the file): mirror of ``lenskit.als.ImplicitMFScorer`` /
``ImplicitMFTrainer`` / ``ALSBase`` / ``ALSTrainerBase`` / ``ALSConfig``
(src/lenskit/als/_common.py:36-356, src/lenskit/als/_implicit.py:24-184), with the same
config fields, attributes after training (``users``, ``items``, ``user_embeddings``,
``item_embeddings``, ``_OtOr``, ``trained_epochs``) and scoring semantics, but with the
epoch loop resident on the GPU (:class:`lkpy_amd._als_engine.ImplicitALSEngine`).
Learned state is kept as host NumPy arrays so pickling / ``get_parameters`` keep working
(SURVEY.md section 5 "Checkpoint / resume"); device state is rebuilt lazily.
"""

from __future__ import annotations

from typing import Literal

import numpy as np
import scipy.sparse as sps
import torch
from pydantic import AliasChoices, BaseModel, Field, PositiveFloat, PositiveInt

from . import _device as D
from . import _native
from ._als_engine import HipBackend, ImplicitALSEngine
from .basic import BiasModel
from .data import Dataset, ItemList, RecQuery, Vocabulary
from .pipeline import Component
from .training import ModelTrainer, TrainingOptions, UsesTrainer


class _DeviceBacked:
    """
    A learned array attribute (``user_embeddings``, ``item_embeddings``, ``_OtOr``) that lives on
    the host -- so that pickling / ``get_parameters`` work as in the reference -- but is only
    REFRESHED FROM HBM WHEN SOMEBODY READS IT: while a trainer is live the factors stay on the
    device across epochs (one PCIe crossing in, one out) and ``train_epoch`` merely marks the
    host copies stale.  The ``ModelTrainer`` contract "the model is usable after every epoch"
    (src/lenskit/training.py:336-378) holds: the first read after an epoch downloads.
    """

    def __set_name__(self, owner, name):
        self.slot = "_h_" + name

    def __get__(self, obj, cls=None):
        if obj is None:
            return self
        pend = obj.__dict__.get("_pending_sync")
        if pend is not None:
            pend()
        return obj.__dict__.get(self.slot)

    def __set__(self, obj, value):
        obj.__dict__[self.slot] = value


def _scorer_state(obj) -> dict:
    "``__getstate__`` of the ALS scorers: host arrays only (synchronised first), no device state."
    pend = obj.__dict__.get("_pending_sync")
    if pend is not None:
        pend()
    st = dict(obj.__dict__)
    st.pop("_dev", None)
    st.pop("_pending_sync", None)
    return st


def _restore_scorer_state(obj, state: dict):
    """
    ``__setstate__`` of the ALS scorers.  Pickles written before the learned arrays became
    lazily-synchronised descriptors hold them under their plain names (``user_embeddings``,
    ``item_embeddings``, ``_OtOr``); they are moved to the descriptors' ``_h_`` slots so that an
    old model file scores as before instead of failing later inside ``_device_state``.
    """
    state = dict(state)
    for name in ("user_embeddings", "item_embeddings", "_OtOr"):
        if name in state and isinstance(getattr(type(obj), name, None), _DeviceBacked):
            state.setdefault("_h_" + name, state.pop(name))
    state.pop("_dev", None)
    state.pop("_pending_sync", None)
    obj.__dict__.update(state)


def _reference_order(options: TrainingOptions) -> str | None:
    """``LK_ALS_RHS_ORDER`` through ``TrainingOptions.environment`` (or the process environment):
    ``auto`` (default: rows of more than 2048 entries summed exactly as the reference sums them),
    ``reference`` (strict: every row of more than 256 entries), ``accurate`` (the tuned kernels'
    own order everywhere) -- INTEGRATION.md, environment knobs; None = not said here, the backend
    reads the process environment."""
    order = options.env_var("LK_ALS_RHS_ORDER", None) if options is not None else None
    return None if not order else order.lower()


class UIPair(BaseModel):
    user: PositiveFloat
    item: PositiveFloat


class ALSConfig(BaseModel):
    "src/lenskit/als/_common.py:36-78 (+ EmbeddingSizeMixin's ``embedding_size_exp``)."

    embedding_size: PositiveInt = Field(
        default=64, validation_alias=AliasChoices("embedding_size", "features"))
    embedding_size_exp: PositiveInt | None = None
    epochs: PositiveInt = 10
    regularization: PositiveFloat | UIPair | dict = 0.1
    user_embeddings: bool | Literal["prefer"] = True

    def model_post_init(self, _ctx):
        if self.embedding_size_exp is not None:
            object.__setattr__(self, "embedding_size", 2 ** int(self.embedding_size_exp))
        if self.embedding_size > 1024:
            # fail at configuration time, not in the middle of training (lk_padded_dim).  Up to
            # 256 the normal matrix stays in registers; 257 .. 1024 take the HBM-tile solver
            # (csrc/als_big.hip: exact, slower)
            raise ValueError(
                f"embedding_size {self.embedding_size} exceeds the device kernels' limit of 1024")
        if isinstance(self.regularization, dict):
            object.__setattr__(self, "regularization", UIPair(**self.regularization))

    @property
    def user_reg(self) -> float:
        r = self.regularization
        return r.user if isinstance(r, UIPair) else float(r)

    @property
    def item_reg(self) -> float:
        r = self.regularization
        return r.item if isinstance(r, UIPair) else float(r)


class ImplicitMFConfig(ALSConfig):
    "src/lenskit/als/_implicit.py:24-32"

    weight: float = 40
    use_ratings: bool = False
    solver: Literal["auto", "cholesky", "cg"] = "auto"
    """Backend knob (also ``LK_ALS_SOLVER``): ``auto`` = ``cholesky`` = the exact solve the
    reference performs (LAPACK sposv), at every supported embedding size (k <= 1024); ``cg`` =
    tolerance-terminated conjugate gradient (64 <= padded k <= 256), on request only."""


_SOLVERS = {"auto": _native.SOLVER_AUTO, "cholesky": _native.SOLVER_CHOLESKY,
            "chol": _native.SOLVER_CHOLESKY, "cg": _native.SOLVER_CG}


class ImplicitMFScorer(UsesTrainer, Component):
    """
    Implicit-feedback matrix factorisation trained with ALS (Hu, Koren, Volinsky); the
    reference solves every row exactly with LAPACK ``sposv`` (class docstring,
    src/lenskit/als/_implicit.py:52-54) and so does the default GPU solver.
    """

    config: ImplicitMFConfig
    accepts_history_batch = True  # recommend_batch takes a lkpy_amd.basic.HistoryBatch

    users: Vocabulary | None = None
    items: Vocabulary
    user_embeddings = _DeviceBacked()  # np.ndarray [users x k] f32 | None
    item_embeddings = _DeviceBacked()  # np.ndarray [items x k] f32
    _OtOr = _DeviceBacked()  # np.ndarray [k x k] f32: Q^T Q + user_reg I

    def create_trainer(self, data, options):
        return ImplicitMFTrainer(self, data, options)

    # -- device-side caches (never pickled) ---------------------------------------
    def __getstate__(self):
        return _scorer_state(self)

    def __setstate__(self, state):
        _restore_scorer_state(self, state)

    def _device_state(self):
        dev = getattr(self, "_dev", None)
        if dev is None or dev["src"] is not self.item_embeddings:
            d = D.device()
            dev = {"src": self.item_embeddings, "device": d,
                   "Q": D.to_device_padded(self.item_embeddings, d),
                   "OtOr": torch.from_numpy(np.ascontiguousarray(self._OtOr)).to(d)}
            self._dev = dev
        return dev

    def _solver(self, options: TrainingOptions | None = None) -> int:
        name = self.config.solver
        if options is not None:
            name = options.env_var("LK_ALS_SOLVER", name) or name
        return _SOLVERS[name.lower()]

    # -- fold-in --------------------------------------------------------------------
    def _history_rows(self, queries: list[RecQuery]):
        """
        histories -> CSR (queries x items) of confidence values (_implicit.py:77-99).
        History items the model does not know are DROPPED: the reference builds the ``ri_good``
        mask for exactly that (_implicit.py:82-90) although its ``numbers()`` call raises
        ``KeyError`` first (default ``missing="error"``, data/_items.py:617,654-655); a batch
        must not fail because one history mentions a new item.
        """
        idx, val, ptr = [], [], [0]
        for q in queries:
            hist = q.query_items
            if hist is not None and len(hist) > 0:
                ri = hist.numbers(vocabulary=self.items, missing="negative")
                good = ri >= 0
                if self.config.use_ratings:
                    ratings = hist.field("rating")
                    if ratings is None:
                        raise ValueError("no ratings in user items")
                    v = np.asarray(ratings)[good] * self.config.weight
                else:
                    v = np.full(int(good.sum()), self.config.weight)
                order = np.argsort(ri[good], kind="stable")
                idx.append(ri[good][order])
                val.append(np.asarray(v, dtype=np.float32)[order])
            ptr.append(ptr[-1] + (len(idx[-1]) if hist is not None and len(hist) > 0 else 0))
        indices = np.concatenate(idx).astype(np.int32) if idx else np.zeros(0, np.int32)
        values = np.concatenate(val).astype(np.float32) if val else np.zeros(0, np.float32)
        return np.asarray(ptr, dtype=np.int64), indices, values

    def new_user_embedding(self, user_num, user_items: ItemList):
        "One fold-in (_implicit.py:77-99); returns (vector, None)."
        st = self._device_state()
        ptr, idx, val = self._history_rows([RecQuery(user_items=user_items)])
        hist = D.DeviceCSR.from_arrays(ptr, idx, val, (1, len(self.items)), st["device"])
        u = D.fold_in(hist, st["Q"], st["OtOr"], self.config.embedding_size, self._solver())
        return D.to_host_unpadded(u, self.config.embedding_size)[0], None

    def _query_embeddings(self, queries: list[RecQuery]) -> tuple[torch.Tensor, np.ndarray]:
        """
        Device [B x KP] embedding per query + validity mask, with ``ALSBase.__call__``'s
        precedence (_common.py:139-157): history present and user_embeddings != "prefer" ->
        fold-in; else the stored row; else invalid.
        """
        st = self._device_state()
        k = self.config.embedding_size
        B = len(queries)
        fold = [q.query_items is not None and len(q.query_items) > 0 and
                self.config.user_embeddings != "prefer" for q in queries]
        out = torch.zeros((B, D.padded_dim(k)), dtype=torch.float32, device=st["device"])
        valid = np.zeros(B, dtype=bool)
        fi = [i for i in range(B) if fold[i]]
        if fi:
            ptr, idx, val = self._history_rows([queries[i] for i in fi])
            hist = D.DeviceCSR.from_arrays(ptr, idx, val, (len(fi), len(self.items)),
                                           st["device"])
            u = D.fold_in(hist, st["Q"], st["OtOr"], k, self._solver())
            out[torch.as_tensor(fi, device=st["device"])] = u
            valid[fi] = True
        rest = [i for i in range(B) if not fold[i]]
        if rest and self.user_embeddings is not None and self.users is not None:
            nums = [None if queries[i].user_id is None else
                    self.users.number(queries[i].user_id, missing=None) for i in rest]
            have = [(i, n) for i, n in zip(rest, nums) if n is not None]
            if have:
                rows = self.user_embeddings[[n for _, n in have]]
                out[torch.as_tensor([i for i, _ in have], device=st["device"])] = \
                    D.to_device_padded(rows, st["device"])
                valid[[i for i, _ in have]] = True
        return out, valid

    # -- scoring (src/lenskit/als/_common.py:133-175) -----------------------------------
    def __call__(self, query, items: ItemList) -> ItemList:
        query = RecQuery.create(query)
        u, valid = self._query_embeddings([query])
        if not valid[0]:
            return ItemList(items, scores=np.nan)
        st = self._device_state()
        all_scores = D.score_dense(u, st["Q"], self.config.embedding_size)[0].cpu().numpy()
        item_nums = items.numbers(vocabulary=self.items, missing="negative")
        mask = item_nums >= 0
        scores = np.full(len(items), np.nan, dtype=np.float32)
        scores[mask] = all_scores[item_nums[mask]]
        return ItemList(items, scores=scores)

    def _history_batch_embeddings(self, batch, pending: list | None = None):
        """
        ``_query_embeddings`` for a :class:`lkpy_amd.basic.HistoryBatch`: the histories' CSR is cut
        out of the HBM-resident training matrix by one kernel (no per-query Python), the fold-in
        is ONE half-epoch launch over it (_implicit.py:77-130 per query in the reference), and
        the precedence of ``ALSBase.__call__`` (_common.py:139-157) is applied with host masks
        over the batch: history present and user_embeddings != "prefer" -> fold-in; else the
        stored row of a known user; else invalid.  Returns (device [B x KP], valid, history CSR).
        """
        st = self._device_state()
        cfg = self.config
        k = cfg.embedding_size
        B = len(batch)
        has_hist = batch.lengths > 0
        stored = np.full(B, -1, dtype=np.int64)
        if self.user_embeddings is not None and self.users is not None:
            if batch.users is self.users or batch.users == self.users:
                stored = batch.user_nums.astype(np.int64)
            else:
                stored = self.users.numbers(batch.user_ids, missing="negative").astype(np.int64)
        fold = has_hist if cfg.user_embeddings != "prefer" else (has_hist & (stored < 0))
        hist = batch.csr(use_ratings=cfg.use_ratings, scale=cfg.weight)
        if fold.all():
            u = D.fold_in(hist, st["Q"], st["OtOr"], k, self._solver(), pending)
        elif fold.any():
            sub = batch.subset(fold).csr(use_ratings=cfg.use_ratings, scale=cfg.weight)
            u = torch.zeros((B, D.padded_dim(k)), dtype=torch.float32, device=st["device"])
            u[torch.from_numpy(np.flatnonzero(fold)).to(st["device"])] = \
                D.fold_in(sub, st["Q"], st["OtOr"], k, self._solver(), pending)
        else:
            u = torch.zeros((B, D.padded_dim(k)), dtype=torch.float32, device=st["device"])
        take = ~fold & (stored >= 0)
        if take.any():
            rows = np.ascontiguousarray(self.user_embeddings[stored[take]], dtype=np.float32)
            u[torch.from_numpy(np.flatnonzero(take)).to(st["device"])] = \
                D.to_device_padded(rows, st["device"])
        return u, fold | take, hist

    def recommend_batch(self, queries, n: int, *, exclude_history: bool = True,
                        device_output: bool = False):
        """
        Batched fold-in + dense scoring + top-N for many queries at once (the reference
        loops queries in Python, src/lenskit/batch/_runner.py:283-308).  ``queries``: a list of
        queries, or a :class:`lkpy_amd.basic.HistoryBatch` (training histories by user number: the
        whole call then has no per-query host work).  Returns (item numbers [B x n] with -1
        padding, scores [B x n] with NaN padding) as host arrays (``device_output``: as device
        tensors, nothing downloaded).
        """
        from .basic import HistoryBatch

        if isinstance(queries, HistoryBatch) and not (
                queries.items is self.items or queries.items == self.items):
            queries = queries.queries()  # (another item vocabulary: the per-query mapping)
        if isinstance(queries, HistoryBatch):
            pending: list = []  # the fold-in's status is read once the scoring is queued behind it
            u, valid, hist = self._history_batch_embeddings(queries, pending)
            st = self._device_state()
            if exclude_history:
                idx, sc = D.score_topk(u, st["Q"], self.config.embedding_size, n, hist.indptr,
                                       hist.indices)
            else:
                idx, sc = D.score_topk(u, st["Q"], self.config.embedding_size, n)
            for plan in pending:
                plan.check_status()  # RuntimeError("ALS solve error: ...") like the fold-in alone
            if not valid.all():
                bad = torch.from_numpy(np.flatnonzero(~valid)).to(st["device"])
                idx[bad] = -1
                sc[bad] = float("nan")
            if device_output:
                return idx, sc
            both = torch.cat([idx.view(torch.float32), sc], dim=1)  # one crossing, not two
            host = D.to_host(both)
            cols = idx.shape[1]
            return host[:, :cols].view(np.int32), host[:, cols:]
        queries = [RecQuery.create(q) for q in queries]
        u, valid = self._query_embeddings(queries)
        st = self._device_state()
        excl_ptr = excl_idx = None
        if exclude_history:
            ptr, idx, _ = self._history_rows(queries)
            excl_ptr = torch.from_numpy(ptr).to(st["device"])
            excl_idx = torch.from_numpy(idx).to(st["device"])
        idx, sc = D.score_topk(u, st["Q"], self.config.embedding_size, n, excl_ptr, excl_idx)
        idx, sc = idx.cpu().numpy(), sc.cpu().numpy()
        idx[~valid] = -1
        sc[~valid] = np.nan
        return idx, sc


class ImplicitMFTrainer(ModelTrainer):
    "``ALSTrainerBase`` + ``ImplicitMFTrainer`` (_common.py:195-356, _implicit.py:133-175)."

    def __init__(self, scorer: ImplicitMFScorer, data: Dataset, options: TrainingOptions):
        self.scorer = scorer
        cfg = scorer.config
        scorer.users, scorer.items = data.users, data.items
        self.rng = options.random_generator()
        import threading

        k = cfg.embedding_size
        init = {}

        def draw():
            # item matrix FIRST, then users, same generator (_common.py:287-301).  NumPy draws
            # outside the GIL: the 14 M normals of an ML-25M model (0.1 s) are drawn while the
            # main thread uploads the matrix and builds both orientations and the plans in HBM
            init["Q"] = self.initial_params(data.item_count, k)
            init["P"] = self.initial_params(data.user_count, k)

        th = threading.Thread(target=draw)
        th.start()
        try:
            ui = self.prepare_matrix(data)
            dev = D.device(None if options.configured_device() in ("cuda", "cpu") else
                           options.configured_device())
            backend = HipBackend(k, dev, scorer._solver(options), _reference_order(options))
            self.engine = ImplicitALSEngine(sps.csr_array(ui), k, cfg.user_reg, cfg.item_reg,
                                            None, None, backend, defer_init=True)
        finally:
            th.join()
        scorer.item_embeddings, scorer.user_embeddings = init["Q"], init["P"]
        self.engine.set_initial(scorer.user_embeddings, scorer.item_embeddings)
        self.epochs_trained = 0

    def prepare_matrix(self, data: Dataset) -> sps.csr_array:
        """
        _implicit.py:141-149 + the ``from_scipy(ui_rates)`` of _common.py:218: confidence values
        ``weight`` (or ``weight * rating``) in CSR.  The reference goes through COO and lets
        SciPy convert; the dataset already holds (user, item)-sorted interactions, so the CSR
        arrays are taken as they are -- same matrix, no host sort -- unless the dataset reports
        repeated pairs, which are summed as SciPy's conversion would.
        """
        ints = data.interactions().matrix()
        rmat = ints.scipy(attribute="rating", layout="csr") if self.scorer.config.use_ratings \
            else ints.scipy(layout="csr")
        vals = np.require(rmat.data, dtype=np.float32) * np.float32(self.scorer.config.weight)
        if getattr(data, "has_duplicates", True):
            # repeated (user, item) pairs: ONE entry with the summed confidence, exactly what
            # the reference's COO -> CSR conversion produces (y gets (2v + 1) once, not (v + 1)
            # twice).  ``scipy()`` hands out the Dataset's OWN index arrays without a copy and
            # ``sum_duplicates`` rewrites indices / indptr in place: canonicalise private copies,
            # never the dataset (tests/test_host_logic.py::test_prepare_matrix_leaves_dataset_alone)
            out = sps.csr_array((vals, rmat.indices.copy(), rmat.indptr.copy()), shape=rmat.shape)
            out.sum_duplicates()
            return out
        return sps.csr_array((vals, rmat.indices, rmat.indptr), shape=rmat.shape)

    def initial_params(self, nrows: int, ncols: int) -> np.ndarray:
        "_implicit.py:152-155"
        mat = self.rng.standard_normal((nrows, ncols), dtype=np.float32) * 0.01
        mat *= mat
        return mat

    def train_epoch(self):
        du, di = self.engine.train_epoch()
        self.engine.check()  # RuntimeError("ALS solve error: ...") like implicit.rs:79
        self.epochs_trained += 1
        # the factors stay in HBM; the host copies are refreshed on first read (_DeviceBacked)
        self.scorer.__dict__["_pending_sync"] = self._sync
        return {"deltaP": float(du.item()), "deltaQ": float(di.item())}

    def _sync(self):
        "download the current factors (called lazily through the scorer's attributes)"
        s = self.scorer
        s.__dict__.pop("_pending_sync", None)
        s.user_embeddings = self.engine.user_embeddings()
        s.item_embeddings = self.engine.item_embeddings()
        s._OtOr = self.engine.otor()  # _save_user_otor (_implicit.py:171-175)

    def finalize(self):
        self._sync()
        if not self.scorer.config.user_embeddings:  # _common.py:318-325
            self.scorer.user_embeddings = None
            self.scorer.users = None

    def get_parameters(self):
        return {"user_embeddings": self.scorer.user_embeddings,
                "item_embeddings": self.scorer.item_embeddings}


# ---------------------------------------------------------------------------------------
# Explicit feedback: biased matrix factorisation
# ---------------------------------------------------------------------------------------


class BiasedMFConfig(ALSConfig):
    "src/lenskit/als/_explicit.py:25-29"

    damping: float | tuple[float, float] | dict[str, float] = 5.0


class BiasedMFScorer(UsesTrainer, Component):
    """
    Biased matrix factorisation trained with ALS on bias-normalised ratings
    (``BiasedMFScorer``, src/lenskit/als/_explicit.py:32-90): the bias model stays on the
    host, the row solves (training and fold-in) run in the explicit mode of the HIP kernel.
    Scoring follows ``ALSBase.__call__`` (src/lenskit/als/_common.py:133-175) +
    ``finalize_scores`` (adds b_g + b_i + b_u back).
    """

    config: BiasedMFConfig

    users: Vocabulary | None = None
    items: Vocabulary
    user_embeddings = _DeviceBacked()
    item_embeddings = _DeviceBacked()
    bias: BiasModel

    def create_trainer(self, data, options):
        return BiasedMFTrainer(self, data, options)

    def __getstate__(self):
        return _scorer_state(self)

    def __setstate__(self, state):
        _restore_scorer_state(self, state)

    def _device_state(self):
        dev = getattr(self, "_dev", None)
        if dev is None or dev["src"] is not self.item_embeddings:
            d = D.device()
            dev = {"src": self.item_embeddings, "device": d,
                   "Q": D.to_device_padded(self.item_embeddings, d)}
            self._dev = dev
        return dev

    def new_user_embedding(self, user_num, user_items: ItemList):
        "_explicit.py:56-74: normalise the ratings with the bias model, one explicit row solve."
        inums = user_items.numbers(vocabulary=self.items, missing="negative")
        ratings = user_items.field("rating")
        assert ratings is not None
        ratings = np.asarray(ratings, dtype=np.float32)
        mask = (inums >= 0) & np.isfinite(ratings)
        biases, u_bias = self.bias.compute_for_items(user_items, None, user_items)
        resid = (ratings - biases)[mask]
        order = np.argsort(inums[mask], kind="stable")
        st = self._device_state()
        hist = D.DeviceCSR.from_arrays(
            np.array([0, int(mask.sum())], dtype=np.int64), inums[mask][order].astype(np.int32),
            resid[order].astype(np.float32), (1, len(self.items)), st["device"])
        u = D.fold_in_explicit(hist, st["Q"], self.config.user_reg, self.config.embedding_size)
        return D.to_host_unpadded(u, self.config.embedding_size)[0], u_bias

    def finalize_scores(self, user_num, items: ItemList, user_bias) -> ItemList:
        "_explicit.py:76-93"
        scores = items.scores()
        if user_bias is None:
            if user_num is not None and self.bias.user_biases is not None:
                user_bias = self.bias.user_biases[user_num]
            else:
                user_bias = 0.0
        biases = self.bias.compute_for_items(items, bias=user_bias)
        return ItemList(items, scores=scores + biases)

    def __call__(self, query, items: ItemList) -> ItemList:
        query = RecQuery.create(query)
        user_num = None
        if query.user_id is not None and self.users is not None:
            user_num = self.users.number(query.user_id, missing=None)
        u_feat, u_off = None, None
        hist = query.query_items
        if hist is not None and len(hist) > 0 and self.config.user_embeddings != "prefer":
            u_feat, u_off = self.new_user_embedding(user_num, hist)
        if u_feat is None:
            if user_num is None or self.user_embeddings is None:
                return ItemList(items, scores=np.nan)
            u_feat = self.user_embeddings[user_num, :]
        st = self._device_state()
        k = self.config.embedding_size
        u = D.to_device_padded(np.ascontiguousarray(u_feat, dtype=np.float32)[None, :],
                               st["device"])
        all_scores = D.score_dense(u, st["Q"], k)[0].cpu().numpy()
        item_nums = items.numbers(vocabulary=self.items, missing="negative")
        mask = item_nums >= 0
        scores = np.full(len(items), np.nan, dtype=np.float32)
        scores[mask] = all_scores[item_nums[mask]]
        return self.finalize_scores(user_num, ItemList(items, scores=scores), u_off)


class BiasedMFTrainer(ModelTrainer):
    "``ALSTrainerBase`` + ``BiasedMFTrainer`` (_common.py:195-356, _explicit.py:93-118)."

    def __init__(self, scorer: BiasedMFScorer, data: Dataset, options: TrainingOptions):
        self.scorer = scorer
        cfg = scorer.config
        scorer.users, scorer.items = data.users, data.items
        self.rng = options.random_generator()
        ui = self.prepare_matrix(data)
        k = cfg.embedding_size
        # item matrix FIRST, then users, same generator (_common.py:287-301)
        scorer.item_embeddings = self.initial_params(data.item_count, k)
        scorer.user_embeddings = self.initial_params(data.user_count, k)
        dev = D.device(None if options.configured_device() in ("cuda", "cpu") else
                       options.configured_device())
        backend = HipBackend(k, dev, _native.SOLVER_CHOLESKY, _reference_order(options))
        self.engine = ImplicitALSEngine(sps.csr_array(ui), k, cfg.user_reg, cfg.item_reg,
                                        scorer.user_embeddings, scorer.item_embeddings, backend,
                                        explicit=True)
        self.epochs_trained = 0

    def prepare_matrix(self, data: Dataset) -> sps.coo_array:
        "_explicit.py:95-102: ratings minus the learned biases, float32"
        rmat = data.interactions().matrix().scipy(attribute="rating", layout="coo")
        self.scorer.bias = BiasModel.learn(data, damping=self.scorer.config.damping)
        return self.scorer.bias.transform_matrix(rmat).astype(np.float32)

    def initial_params(self, nrows: int, ncols: int) -> np.ndarray:
        "_explicit.py:104-108: N(0,1) rows scaled to unit length"
        mat = self.rng.standard_normal((nrows, ncols), dtype=np.float32)
        mat /= np.linalg.norm(mat, axis=1).reshape((nrows, 1))
        return mat

    def train_epoch(self):
        du, di = self.engine.train_epoch()
        self.engine.check()  # RuntimeError("ALS solve error: ...") like explicit.rs:72
        self.epochs_trained += 1
        self.scorer.__dict__["_pending_sync"] = self._sync  # lazy download (_DeviceBacked)
        return {"deltaP": float(du.item()), "deltaQ": float(di.item())}

    def _sync(self):
        s = self.scorer
        s.__dict__.pop("_pending_sync", None)
        s.user_embeddings = self.engine.user_embeddings()
        s.item_embeddings = self.engine.item_embeddings()

    def finalize(self):
        self._sync()
        if not self.scorer.config.user_embeddings:  # _common.py:318-325
            self.scorer.user_embeddings = None
            self.scorer.users = None

    def get_parameters(self):
        return {"user_embeddings": self.scorer.user_embeddings,
                "item_embeddings": self.scorer.item_embeddings}
