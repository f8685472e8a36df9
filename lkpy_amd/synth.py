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
    """
    Log-normal activity clipped to [lo, hi], rescaled so that the lengths sum to EXACTLY nnz,
    with the most active user at exactly ``hi`` (ML-25M: 32 202).
    """
    raw = rng.lognormal(mean=0.0, sigma=1.25, size=n_users)
    lens = lo + raw * (nnz / n_users - lo) / raw.mean()
    lens[np.argmax(lens)] = hi  # the longest row is pinned
    for _ in range(12):  # clip + renormalise the unclipped mass
        lens = np.clip(lens, lo, hi)
        free = (lens > lo) & (lens < hi)
        excess = lens.sum() - nnz
        if abs(excess) < 1 or not free.any():
            break
        lens[free] -= excess * (lens[free] - lo) / (lens[free] - lo).sum()
    lens = np.clip(np.rint(lens), lo, hi).astype(np.int64)
    # exact total: +-1 on random rows strictly inside the range
    diff = int(nnz - lens.sum())
    while diff != 0:
        step = 1 if diff > 0 else -1
        ok = np.flatnonzero((lens < hi - 1) & (lens > lo + 1))
        pick = rng.choice(ok, min(abs(diff), len(ok)), replace=False)
        lens[pick] += step
        diff = int(nnz - lens.sum())
    return lens


def _capped_zipf_weights(n_rated, nnz, cap):
    "expected item degrees ~ 1/rank, capped at ``cap``, summing to nnz; returned as probabilities"
    r = np.arange(1, n_rated + 1, dtype=np.float64)
    lo, hi = 1.0, float(nnz)
    for _ in range(80):  # bisection on the scale c of min(c / r, cap)
        c = 0.5 * (lo + hi)
        if np.minimum(c / r, cap).sum() > nnz:
            hi = c
        else:
            lo = c
    w = np.minimum(0.5 * (lo + hi) / r, cap)
    return w / w.sum()


def ml25m_like(seed: int = 20260925, scale: float = 1.0, **overrides) -> sps.csr_array:
    """
    Users x items CSR of ratings (float32), rows sorted by item, no duplicates.  At
    ``scale = 1`` the public dataset's statistics are met EXACTLY:
    nnz = 25 000 095, longest user row 32 202 (shortest 20), busiest item 81 491, 3 376 of the
    62 423 items unrated.  ``scale`` < 1 shrinks users, items and nnz together (for tests).
    """
    cfg = dict(ML25M)
    cfg.update(overrides)
    n_users = max(8, int(cfg["n_users"] * scale))
    n_items = max(16, int(cfg["n_items"] * scale))
    n_empty = int(cfg["n_empty_items"] * scale)
    nnz = int(cfg["nnz"] * scale)
    n_rated = n_items - n_empty
    lo = min(cfg["min_user"], max(1, n_rated // 4))
    # the longest row: the dataset's own maximum when the (scaled) catalogue can hold it
    hi = cfg["max_user"] if cfg["max_user"] <= n_rated * 9 // 10 else n_rated // 2
    cap = max(int(cfg["max_item"] * scale), 2)
    cap = min(cap, n_users - 1)
    rng = np.random.default_rng(seed)

    lens = _user_lengths(rng, n_users, nnz, lo, hi)
    w = _capped_zipf_weights(n_rated, nnz, cap)
    item_of_rank = rng.permutation(n_items)[:n_rated]  # popularity rank -> item id
    cdf = np.cumsum(w)
    cdf[-1] = 1.0

    # light users: oversample WITH replacement, dedupe; heavy users (and any light user the
    # dedupe left short): exact weighted sampling WITHOUT replacement (exponential race:
    # the d smallest of E_i / w_i)
    heavy = lens > max(2000, n_rated // 16)
    light_ids = np.flatnonzero(~heavy)
    over = (lens[light_ids] * 1.7).astype(np.int64) + 24
    uid = np.repeat(light_ids, over)
    draws = np.searchsorted(cdf, rng.random(uid.shape[0]), side="right")
    key = np.unique(uid * n_rated + np.minimum(draws, n_rated - 1))  # (user, rank), sorted, distinct
    uid, rk = key // n_rated, key % n_rated
    cnt = np.bincount(uid, minlength=n_users)
    short = np.flatnonzero((cnt < lens) & ~heavy)
    if len(short):  # drop their partial rows; they are redrawn exactly below
        keep = ~np.isin(uid, short)
        uid, rk = uid[keep], rk[keep]
        cnt = np.bincount(uid, minlength=n_users)
    # trim light rows to their target lengths (random subset)
    start = np.zeros(n_users + 1, dtype=np.int64)
    np.cumsum(cnt, out=start[1:])
    order = np.lexsort((rng.random(uid.shape[0]), uid))
    pos = np.arange(uid.shape[0]) - np.repeat(start[:-1], cnt)
    keep = np.zeros(uid.shape[0], dtype=bool)
    keep[order[pos < np.repeat(np.minimum(lens, cnt), cnt)]] = True
    uid, rk = uid[keep], rk[keep]
    exact_ids = np.concatenate([np.flatnonzero(heavy), short])
    ex_u, ex_r = [], []
    for u in exact_ids:
        d = int(lens[u])
        race = rng.exponential(size=n_rated) / w
        pick = np.argpartition(race, d - 1)[:d]
        ex_u.append(np.full(d, u, dtype=np.int64))
        ex_r.append(pick.astype(np.int64))
    if ex_u:
        uid = np.concatenate([uid] + ex_u)
        rk = np.concatenate([rk] + ex_r)
    # item cap: no item above `cap`, the busiest one exactly at it (full scale: 81 491).
    key = np.sort(uid * n_rated + rk)  # (user, rank) pairs, distinct by construction
    uid, rk = key // n_rated, key % n_rated
    deg = np.bincount(rk, minlength=n_rated)
    over = np.flatnonzero(deg > cap)
    if len(over):
        # surplus raters of an over-full item swap it for a random item of the uncrowded
        # middle of the catalogue that they do not have yet (a few rounds of redraws)
        remove = np.zeros(len(key), dtype=bool)
        for r in over:
            idx = np.flatnonzero(rk == r)
            remove[rng.choice(idx, int(deg[r] - cap), replace=False)] = True
        pending = uid[remove]
        key = key[~remove]
        band = np.flatnonzero(deg < cap // 2)
        band = band[band > over.max()]
        while len(pending):
            newkey = pending * n_rated + rng.choice(band, len(pending))
            uniq, first = np.unique(newkey, return_index=True)
            pos = np.minimum(np.searchsorted(key, uniq), len(key) - 1)
            fresh = key[pos] != uniq
            key = np.sort(np.concatenate([key, uniq[fresh]]))
            done = np.zeros(len(pending), dtype=bool)
            done[first[fresh]] = True
            pending = pending[~done]
        uid, rk = key // n_rated, key % n_rated
        deg = np.bincount(rk, minlength=n_rated)
    if deg[0] < cap:
        # raise the top item to exactly the cap: users lacking it trade in their rarest item
        # (rows are sorted by rank, so that is the row's last entry; it is not rank 0)
        has0 = np.zeros(n_users, dtype=bool)
        has0[uid[rk == 0]] = True
        lacking = rng.permutation(np.flatnonzero(~has0))[: int(cap - deg[0])]
        last = np.zeros(n_users + 1, dtype=np.int64)
        np.cumsum(np.bincount(uid, minlength=n_users), out=last[1:])
        rk[last[lacking + 1] - 1] = 0
    items = item_of_rank[rk].astype(np.int64)
    order = np.lexsort((items, uid))
    uid, items = uid[order], items[order]
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
