"""
Component seam for item-based k-NN: mirror of ``lenskit.knn.ItemKNNScorer`` /
``ItemKNNConfig`` (src/lenskit/knn/item.py:41-295).  Matrix preparation is the reference's
own SciPy code path (item-mean centring, L2 normalisation); the similarity build and the
scoring run in the HIP kernels.
"""

from __future__ import annotations

import warnings
from typing import Literal

import numpy as np
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
        # Arrow extension array, int64 offsets (item.py:176-177: LargeList -> from_array)
        self.sim_matrix = SparseRowArray.from_arrays(
            offsets, out.indices.cpu().numpy(), out.values.cpu().numpy(),
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
