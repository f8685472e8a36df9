"""
Seeded synthetic interaction data shaped like the benchmark datasets.

There is no network and no MovieLens-25M on the bench box, so ``bench.py`` and the
full-size property tests use a seeded stand-in with the public dataset's shape
(SURVEY.md section 8d): U = 162 541 users, I = 62 423 listed items of which 3 376 have
no rating, nnz = 25 000 095; user activity ~ shifted log-normal clipped to
[20, 32 202]; item popularity ~ Zipf(1.0) capped at 81 491; ratings in {0.5..5.0} with
the ml-latest-small histogram.  Host-side NumPy only (data preparation, like the
reference's dataset loaders) -- nothing here is on the timed path.
"""

from __future__ import annotations

import numpy as np
import scipy.sparse as sps

ML25M = dict(n_users=162_541, n_items=62_423, n_empty_items=3_376, nnz=25_000_095,
             min_user=20, max_user=32_202, max_item=81_491)  # fmt: skip

# rating histogram of ml-latest-small (0.5 .. 5.0 in half-star steps)
_RATING_VALUES = np.arange(1, 11, dtype=np.float32) * 0.5
_RATING_PROBS = np.array([1101, 3326, 1687, 7271, 4449, 20064, 10538, 28750, 7723, 15095], float)
_RATING_PROBS /= _RATING_PROBS.sum()


def _user_lengths(rng, n_users, nnz, lo, hi):
    "Log-normal activity, clipped, rescaled so that the lengths sum to ~nnz."
    raw = rng.lognormal(mean=0.0, sigma=1.25, size=n_users)
    lens = lo + raw * (nnz / n_users - lo) / raw.mean()
    for _ in range(8):  # clip + renormalise the unclipped mass
        lens = np.clip(lens, lo, hi)
        free = (lens > lo) & (lens < hi)
        excess = lens.sum() - nnz
        if abs(excess) < 1 or not free.any():
            break
        lens[free] -= excess * (lens[free] - lo) / (lens[free] - lo).sum()
    return np.clip(np.rint(lens), lo, hi).astype(np.int64)


def ml25m_like(seed: int = 20260925, scale: float = 1.0, **overrides) -> sps.csr_array:
    """
    Users x items CSR of ratings (float32), rows sorted by item, no duplicates.
    ``scale`` < 1 shrinks users, items and nnz together (for tests).
    """
    cfg = dict(ML25M)
    cfg.update(overrides)
    n_users = max(8, int(cfg["n_users"] * scale))
    n_items = max(16, int(cfg["n_items"] * scale))
    n_empty = int(cfg["n_empty_items"] * scale)
    nnz = int(cfg["nnz"] * scale)
    lo = min(cfg["min_user"], max(1, (n_items - n_empty) // 4))
    hi = min(cfg["max_user"], (n_items - n_empty) // 2)
    rng = np.random.default_rng(seed)

    lens = _user_lengths(rng, n_users, nnz, lo, hi)
    # item popularity: Zipf(1.0) over the rated items, capped, shuffled over ids
    n_rated = n_items - n_empty
    w = 1.0 / np.arange(1, n_rated + 1, dtype=np.float64)
    cap = cfg["max_item"] * scale / max(nnz, 1)
    w = np.minimum(w / w.sum(), cap)
    w /= w.sum()
    item_of_rank = rng.permutation(n_items)[:n_rated]
    cdf = np.cumsum(w)
    cdf[-1] = 1.0

    # oversample with replacement, dedupe per user, trim to the target lengths
    over = (lens * 1.6).astype(np.int64) + 16
    uid = np.repeat(np.arange(n_users, dtype=np.int64), over)
    draws = np.searchsorted(cdf, rng.random(uid.shape[0]), side="right")
    items = item_of_rank[np.minimum(draws, n_rated - 1)].astype(np.int64)
    key = np.unique(uid * n_items + items)  # sorted by (user, item), duplicates removed
    uid = key // n_items
    items = key % n_items
    # keep at most lens[u] entries per user (random subset, order restored)
    start = np.searchsorted(uid, np.arange(n_users))
    cnt = np.diff(np.append(start, uid.shape[0]))
    rnk = rng.random(uid.shape[0])
    order = np.lexsort((rnk, uid))
    pos = np.arange(uid.shape[0]) - np.repeat(start, cnt)
    keep = np.zeros(uid.shape[0], dtype=bool)
    keep[order[pos < np.repeat(np.minimum(lens, cnt), cnt)]] = True
    uid, items = uid[keep], items[keep]
    ratings = rng.choice(_RATING_VALUES, size=uid.shape[0], p=_RATING_PROBS).astype(np.float32)
    indptr = np.zeros(n_users + 1, dtype=np.int64)
    np.cumsum(np.bincount(uid, minlength=n_users), out=indptr[1:])
    if indptr[-1] < np.iinfo(np.int32).max:
        indptr = indptr.astype(np.int32)
    return sps.csr_array((ratings, items.astype(np.int32), indptr), shape=(n_users, n_items))


def describe(mat: sps.csr_array) -> dict:
    ul = np.diff(mat.indptr)
    il = np.bincount(mat.indices, minlength=mat.shape[1])
    return {
        "n_users": int(mat.shape[0]),
        "n_items": int(mat.shape[1]),
        "nnz": int(mat.nnz),
        "user_len_min": int(ul.min()),
        "user_len_mean": float(ul.mean()),
        "user_len_max": int(ul.max()),
        "item_len_max": int(il.max()),
        "empty_items": int((il == 0).sum()),
        "sum_user_len_sq": int((ul.astype(np.int64) ** 2).sum()),
    }


# ---------------------------------------------------------------------------------------
# cfg5: 10^7 users x 10^6 items x 10^8 interactions, generated in HBM
# ---------------------------------------------------------------------------------------

CFG5 = dict(n_users=10_000_000, n_items=1_000_000, nnz=100_000_000, seed=5, max_degree=4096)


def zipf_degrees(n_rows: int, nnz: int, seed: int, max_degree: int = 4096) -> np.ndarray:
    """
    Per-user degrees ~ truncated power law P(d) ~ d^-s on [1, max_degree] with the exponent
    solved for the requested mean (10 for cfg5), drawn from ``np.random.Philox(seed)``, then
    nudged by +-1 on random rows so that they sum to ``nnz`` EXACTLY (min 1 kept).
    """
    d = np.arange(1, max_degree + 1, dtype=np.float64)
    target = nnz / n_rows
    lo, hi = 0.5, 4.0
    for _ in range(60):  # bisection on the exponent
        s = 0.5 * (lo + hi)
        w = d ** -s
        if (w * d).sum() / w.sum() > target:
            lo = s
        else:
            hi = s
    w = d ** -(0.5 * (lo + hi))
    cdf = np.cumsum(w / w.sum())
    cdf[-1] = 1.0
    rng = np.random.Generator(np.random.Philox(seed))
    deg = (np.searchsorted(cdf, rng.random(n_rows), side="right") + 1).astype(np.int64)
    deg = np.minimum(deg, max_degree)
    diff = int(nnz - deg.sum())
    while diff != 0:
        step = 1 if diff > 0 else -1
        ok = np.flatnonzero((deg < max_degree) if step > 0 else (deg > 1))
        pick = rng.choice(ok, min(abs(diff), len(ok)), replace=False)
        deg[pick] += step
        diff = int(nnz - deg.sum())
    return deg


def zipf_csr_on_device(dev, n_users: int, n_items: int, nnz: int, seed: int = 5,
                       value: float = 40.0, max_degree: int = 4096):
    """
    The cfg5-style matrix as a :class:`lkpy_amd._device.DeviceCSR` (users x items, values =
    ``value``), generated by ``lk_synth_zipf_rows`` (csrc/synth.hip) -- only the 8-byte-per-user
    offsets cross PCIe.  Rows hold distinct items in ascending order.
    """
    import ctypes

    import torch

    from . import _device as D
    from . import _native

    lib = _native.require_gpu()
    deg = zipf_degrees(n_users, nnz, seed, max_degree)
    indptr = np.zeros(n_users + 1, dtype=np.int64)
    np.cumsum(deg, out=indptr[1:])
    long_rows = np.flatnonzero(deg > 32).astype(np.int32)
    d_ptr = torch.from_numpy(indptr).to(dev)
    d_long = torch.from_numpy(long_rows).to(dev)
    idx = torch.empty(nnz, dtype=torch.int32, device=dev)
    _native.check(lib.lk_synth_zipf_rows(
        D._ptr(d_ptr), n_users, n_items, ctypes.c_uint64(seed), D._ptr(d_long), len(long_rows),
        D._ptr(idx), D._stream()), "lk_synth_zipf_rows")
    vals = torch.full((nnz,), float(value), dtype=torch.float32, device=dev)
    if nnz < np.iinfo(np.int32).max:
        h_ptr = indptr.astype(np.int32)
        d_ptr = d_ptr.to(torch.int32)
    else:
        h_ptr = indptr
    return D.DeviceCSR(d_ptr, idx, vals, (n_users, n_items), h_ptr)
