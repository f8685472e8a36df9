"""Deterministic INPUTS of the reference-generated ALS fixtures (``make_als_fixtures.py``).

Everything here comes from integer hashes (splitmix64 finaliser) evaluated with NumPy
``uint64`` arithmetic -- no dependence on an RNG stream that a NumPy upgrade could change --
so the tests rebuild exactly the inputs the reference functions were run on, and the committed
fixtures only hold the reference's OUTPUTS.

TEST INFRASTRUCTURE: imported by ``tests/`` and by the fixture generator only.
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path

import numpy as np
import scipy.sparse as sps

N_CATALOGUE = 50_000  # rows of the synthetic "other" factor matrix
ROW_K = (25, 64, 128, 256)
ROW_N = (1, 2, 5, 16, 17, 64, 65, 100, 1000, 2048, 2049, 5000, 40_000)
REG = 0.1

# cfg1: ml-latest-small, k = 25, weight 40, reg 0.1, seed 42
ML_K, ML_REG, ML_WEIGHT, ML_SEED = 25, 0.1, 40.0, 42


def _mix(x: np.ndarray) -> np.ndarray:
    "splitmix64 finaliser on uint64 arrays"
    with np.errstate(over="ignore"):
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def hash64(stream: int, n: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        base = _mix(np.asarray([stream], dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))[0]
        return _mix(base + np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))


def uniform(stream: int, n: int) -> np.ndarray:
    "float64 in [0, 1), 53 bits"
    return (hash64(stream, n) >> np.uint64(11)).astype(np.float64) * 2.0**-53


@dataclass(frozen=True)
class RowCase:
    kind: str  # "centered": zero-mean factors, constant confidence; "skewed": positive-mean, ratings x weight
    k: int
    n: int
    reg: float = REG

    @property
    def name(self) -> str:
        return f"{self.kind}_k{self.k}_n{self.n}"


def row_cases():
    return [RowCase(kind, k, n) for kind in ("centered", "skewed") for k in ROW_K for n in ROW_N]


_EMB_CACHE: dict = {}


def embeddings(case: RowCase) -> np.ndarray:
    "the 'other' factor matrix of a case: N_CATALOGUE x k float32"
    key = (case.kind, case.k)
    if key not in _EMB_CACHE:
        u = uniform(1000 + case.k + (0 if case.kind == "centered" else 7), N_CATALOGUE * case.k)
        if case.kind == "centered":
            m = (u - 0.5) * (2.0 / np.sqrt(case.k))
        else:  # what trained implicit-feedback factors look like: mostly positive, small
            m = (u - 0.25) * (0.6 / np.sqrt(case.k))
        _EMB_CACHE.clear()  # one 51 MB matrix at a time
        _EMB_CACHE[key] = np.ascontiguousarray(m.reshape(N_CATALOGUE, case.k), dtype=np.float32)
    return _EMB_CACHE[key]


def row_entries(case: RowCase):
    "(sorted distinct item numbers int32 [n], confidence values float32 [n])"
    stream = 50_000 + case.k * 131 + case.n
    order = np.argsort(hash64(stream, N_CATALOGUE), kind="stable")
    items = np.sort(order[: case.n]).astype(np.int32)
    if case.kind == "centered":
        vals = np.full(case.n, ML_WEIGHT, dtype=np.float32)
    else:
        stars = np.floor(uniform(stream + 1, case.n) * 10.0) * 0.5 + 0.5  # 0.5 ... 5.0
        vals = (stars.astype(np.float32) * np.float32(ML_WEIGHT)).astype(np.float32)
    return items, vals


def explicit_values(case: RowCase) -> np.ndarray:
    "bias-normalised ratings of the explicit model for the row of ``case``: float32 in [-2.5, 2.5)"
    stream = 90_000 + case.k * 131 + case.n
    return ((uniform(stream, case.n) - 0.5) * 5.0).astype(np.float32)


EXPLICIT_N = (1, 2, 5, 16, 17, 64, 65, 100, 1000, 2048, 2049, 5000)  # (40 000: a 40 MB product)


def explicit_cases():
    return [RowCase(kind, k, n) for kind in ("centered", "skewed") for k in ROW_K
            for n in EXPLICIT_N]


EASE_ITEMS, EASE_REG = 1200, 1.0
EASE_ROWS = np.arange(0, EASE_ITEMS, 50)  # the rows of the inverse that are committed


def ease_binary_matrix(path: Path | None = None) -> sps.csr_array:
    "users x (the 1200 most-rated ml-latest-small items), binary, float32"
    ui, _ = ml_small_matrices(path)
    counts = np.diff(sps.csc_array(ui).indptr)
    top = np.sort(np.argsort(-counts, kind="stable")[:EASE_ITEMS])
    x = sps.csr_array(ui[:, top], dtype=np.float32)
    x.data[:] = 1.0
    return x


def ease_cooc(path: Path | None = None) -> np.ndarray:
    "dense item-item co-occurrence counts incl. the diagonal (knn/ease.py:108), float32"
    x = ease_binary_matrix(path)
    return np.asarray((x.T @ x).todense(), dtype=np.float32)


def ml_small_matrices(path: Path | None = None):
    """
    ml-latest-small as the reference's ``ml_ds`` fixture sees it (items = every movies.csv id,
    sorted; users = sorted rating user ids), implicit: values = weight * 1 (``prepare_matrix``,
    ``src/lenskit/als/_implicit.py:141-149``).  Returns (users x items CSR, items x users CSR),
    float32 values, int32 sorted indices.
    """
    if path is None:
        path = Path(__file__).resolve().parent / "ml_small.npz"
    z = np.load(path)
    user_ids = np.unique(z["user_id"])
    item_ids = np.unique(z["all_item_ids"])
    rows = np.searchsorted(user_ids, z["user_id"]).astype(np.int32)
    cols = np.searchsorted(item_ids, z["item_id"]).astype(np.int32)
    vals = np.full(len(rows), ML_WEIGHT, dtype=np.float32)
    coo = sps.coo_array((vals, (rows, cols)), shape=(len(user_ids), len(item_ids)))
    ui = sps.csr_array(coo)
    iu = sps.csr_array(coo.T)
    ui.sort_indices()
    iu.sort_indices()
    return ui, iu
