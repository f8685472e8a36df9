"""
At-scale parity accounting -- TEST INFRASTRUCTURE ONLY (used by ``tests/`` and by the
``parity`` / ``cpu_baseline`` legs of ``bench.py``, never by the product package).

``north_star``: factors within 1e-4 relative of the reference CPU path.  Two float32
implementations of ``x = A^-1 y`` (the reference's ndarray + LAPACK ``sposv`` arithmetic,
restated in ``lk_oracle.c``, and the HIP kernels) can each only be expected within about
``cond(A) * 2^-24`` of the exact solution, so the 1e-4 criterion is decidable only for rows
whose conditioning permits it.  The accounting below therefore reports, per half-epoch run
from IDENTICAL inputs on both sides:

* the matrix-level relative error GPU vs oracle (Frobenius),
* how many rows exceed 1e-4 (row-wise ``||x_gpu - x_oracle|| / ||x_oracle||``) and the
  condition numbers of exactly those rows (lower-bound estimates from the float64 referee),
* the claims that are asserted: EVERY row with ``cond * 2^-24 < 1e-5`` is within 1e-4 of the
  oracle -- or, where it is not, the float64 referee shows that the gap is the REFERENCE
  arithmetic's own (the GPU row is within 0.5e-4 of the exact answer; such rows are listed with
  all three distances: they are rows of tens of thousands of entries whose sequential float32
  accumulation in the reference drifts by ~1e-4); the
  GPU is no further from the float64 referee than the reference arithmetic is (matrix level,
  factor 2 slack); and -- the statement that is not vacuous on real data, where
  ``cond(A) >= 100`` for every non-empty row because ``cond(OtOr)`` already is -- EVERY row of
  the GPU result lies within ``NORM_BOUND * cond * 2^-24 + FLOOR`` of the float64 answer (the a-priori
  forward error of a backward-stable float32 solve, measured constant: 1.3 / 2.4 for the HIP
  kernels on the ML-25M-shaped epoch, 1.3 / 7.3 for the reference arithmetic),
* a histogram of row error against cond (decades), so nothing hides behind a loose bound.
"""

from __future__ import annotations

import numpy as np

U32 = 2.0**-24  # unit roundoff of float32
RTOL = 1.0e-4  # north_star tolerance
COND_LIMIT = 1.0e-5 / U32  # rows with cond below this must meet RTOL (~168)
NORM_BOUND = 4.0  # every GPU row within NORM_BOUND * cond * u + FLOOR of the float64 answer
FLOOR = 1.0e-5


def _row_rel(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    "row-wise ||a - b|| / ||b|| (0 where both are zero rows)"
    num = np.linalg.norm(a.astype(np.float64) - b.astype(np.float64), axis=1)
    den = np.linalg.norm(b.astype(np.float64), axis=1)
    out = np.zeros_like(num)
    nz = den > 0
    out[nz] = num[nz] / den[nz]
    out[~nz & (num > 0)] = np.inf
    return out


def _rel(a, b) -> float:
    d = np.linalg.norm(a.astype(np.float64) - b.astype(np.float64))
    return float(d / max(np.linalg.norm(b.astype(np.float64)), 1e-300))


def als_half_accounting(got: np.ndarray, want: np.ndarray, exact: np.ndarray | None,
                        cond: np.ndarray | None) -> dict:
    """
    ``got``: GPU rows, ``want``: oracle (reference arithmetic) rows, ``exact``: float64 referee
    rows, ``cond``: per-row condition estimates -- all for the same half-epoch from the same
    inputs.  Returns the accounting dict.

    ``ok`` is the RAW north-star criterion and nothing else: no row further than 1e-4 (relative)
    from the oracle's row.  Rows that fail it are listed in ``exceptions`` with their condition
    number and all three distances (GPU-oracle, GPU-float64, oracle-float64).  ``accounted`` is
    the separate, weaker statement used where the data cannot meet the raw criterion
    (ml-latest-small, a first epoch from the tiny init: cond 1e3 ... 1e6): every row over 1e-4 is
    explained by its conditioning or by the reference arithmetic's own distance from float64.
    """
    e_go = _row_rel(got, want)
    over = e_go > RTOL
    res = {
        "rows": int(got.shape[0]),
        "rel_gpu_vs_oracle": _rel(got, want),
        "row_rel_max": float(e_go.max()) if len(e_go) else 0.0,
        "row_rel_p50": float(np.median(e_go)) if len(e_go) else 0.0,
        "row_rel_p999": float(np.quantile(e_go, 0.999)) if len(e_go) else 0.0,
        "rows_over_1e-4": int(over.sum()),
    }
    res["ok"] = bool(not over.any())
    ok = True
    if exact is not None:
        res["rel_gpu_vs_f64"] = _rel(got, exact)
        res["rel_oracle_vs_f64"] = _rel(want, exact)
        # at least as accurate as the reference arithmetic (factor 2 + 1e-6 slack)
        ok &= res["rel_gpu_vs_f64"] <= 2.0 * res["rel_oracle_vs_f64"] + 1.0e-6
    if cond is not None:
        cu = cond * U32
        decidable = cu < 1.0e-5
        res["rows_decidable"] = int(decidable.sum())  # cond * 2^-24 < 1e-5
        res["decidable_rows_over_1e-4"] = int((over & decidable).sum())
        if exact is None:
            ok &= res["decidable_rows_over_1e-4"] == 0
        else:
            # a decidable row over 1e-4 counts against the GPU only if the GPU row itself is
            # more than half the tolerance from the float64 answer; otherwise the gap is the
            # reference arithmetic's own deviation, and the row is listed with all distances
            e_gx = _row_rel(got, exact)
            e_ox = _row_rel(want, exact)
            mine = over & decidable & (e_gx > 0.5 * RTOL)
            res["decidable_rows_over_1e-4_gpu_side"] = int(mine.sum())
            ok &= not mine.any()
            res["exceptions"] = [
                {"row": int(r), "cond": float(cond[r]), "gpu_vs_oracle": float(e_go[r]),
                 "gpu_vs_f64": float(e_gx[r]), "oracle_vs_f64": float(e_ox[r])}
                for r in np.flatnonzero(over)[:10]]
        if over.any():
            res["min_cond_of_rows_over"] = float(cond[over].min())
            res["max_cond_of_rows_over"] = float(cond[over].max())
        res["max_cond"] = float(cond.max()) if len(cond) else 0.0
        # histogram: error normalised by cond * u, by decade of cond
        nzc = cond > 0
        hist = {}
        if nzc.any():
            dec = np.floor(np.log10(np.maximum(cond[nzc], 1.0))).astype(int)
            for d in np.unique(dec):
                m = dec == d
                hist["1e%d" % d] = {
                    "rows": int(m.sum()),
                    "row_rel_max": float(e_go[nzc][m].max()),
                    "row_rel_over_cond_u_max": float((e_go[nzc][m] / cu[nzc][m]).max()),
                }
        res["by_cond_decade"] = hist
        if exact is not None:
            e_g = _row_rel(got, exact)
            e_o = _row_rel(want, exact)
            res["row_err_over_cond_u_max_gpu"] = float((e_g[nzc] / cu[nzc]).max()) if nzc.any() else 0.0
            res["row_err_over_cond_u_max_oracle"] = float((e_o[nzc] / cu[nzc]).max()) if nzc.any() else 0.0
            # cond is a LOWER-bound estimate, which only makes this test stricter; FLOOR covers
            # what no condition number explains: forming A and y is itself a k- and n-term
            # float32 accumulation (~ k * u ~ 1e-5 at k = 256), ten times below the tolerance
            # A row that is within the tolerance of the ORACLE is what the reference computes,
            # however far both are from float64: since round 5 the default plans evaluate rows of
            # more than 2048 entries in the reference's own summation order and reproduce its
            # drift there (19 x cond u on the busiest ML-25M items at k = 256) -- only rows that
            # miss the oracle AND the forward bound are unexplained.
            beyond = e_g[nzc] > NORM_BOUND * cu[nzc] + FLOOR
            viol = beyond & (e_go[nzc] > RTOL)
            res["rows_beyond_forward_bound"] = int(beyond.sum())
            res["rows_beyond_forward_bound_and_over_1e-4"] = int(viol.sum())
            ok &= not viol.any()
    res["accounted"] = bool(ok)
    return res


def knn_rows_equal(gpu_ptr, gpu_idx, gpu_val, want) -> dict:
    """
    Bitwise comparison of sampled similarity rows: ``gpu_*`` is a CSR (offsets, int32 columns,
    f32 values) of the sampled rows as the GPU produced them, ``want`` the oracle's
    ``iknn_build_rows`` result for the same rows.
    """
    wp = np.asarray(want.indptr, dtype=np.int64)
    gp = np.asarray(gpu_ptr, dtype=np.int64)
    same_ptr = bool(np.array_equal(gp - gp[0], wp))
    same_idx = same_ptr and bool(np.array_equal(np.asarray(gpu_idx, np.int32), want.indices))
    same_val = same_idx and bool(
        np.array_equal(np.asarray(gpu_val, np.float32).view(np.uint32),
                       np.asarray(want.data, np.float32).view(np.uint32)))
    return {"rows_checked": int(len(wp) - 1), "entries_checked": int(wp[-1]),
            "bitwise_equal": bool(same_ptr and same_idx and same_val)}


def topn_accounting(got_i, got_s, want_i, want_s, dense_scores_of_row) -> dict:
    """
    Top-N lists of the GPU (``got_*``) against the oracle's (``want_*``) for the same queries
    FROM THE SAME QUERY VECTORS, [rows x n] with -1 / NaN padding.  ``north_star``: "integer
    top-K index sets bit-exact".  The sorted score rows must be bit-identical position by
    position and every listed item must really carry the listed score.  Where DIFFERENT items
    have the same score bits (duplicate factor rows: items with identical interaction patterns)
    the reference's heap decides by its internal sift order -- which of two equal scores it pops
    first inside the list, and, when the tie straddles the cut, which one it keeps
    (src/accel/indirect/heap.rs:39-64 compares scores only; SURVEY.md section 8g item 8:
    unspecified); the GPU takes the lower item number.  Such rows are counted in the two ``ties_*``
    fields, anything else in ``mismatched_users``.  ``dense_scores_of_row(r)`` -> the oracle's
    score of every item for row ``r`` (only called for rows whose lists differ).
    """
    got_i, want_i = np.asarray(got_i), np.asarray(want_i)
    got_s = np.ascontiguousarray(got_s, dtype=np.float32)
    want_s = np.ascontiguousarray(want_s, dtype=np.float32)
    same = (got_i == want_i).all(axis=1)
    sc_same = bool(np.array_equal(got_s.view(np.uint32), want_s.view(np.uint32)))
    ties_in, ties_cut, bad = 0, 0, 0
    for r in np.flatnonzero(~same):
        sc = dense_scores_of_row(int(r))
        g, w = got_i[r], want_i[r]
        genuine = np.array_equal(sc[g[g >= 0]].view(np.uint32),
                                 got_s[r][g >= 0].view(np.uint32))
        rows_equal = np.array_equal(got_s[r].view(np.uint32), want_s[r].view(np.uint32))
        if not (genuine and rows_equal):
            bad += 1
        elif np.array_equal(np.sort(g), np.sort(w)):
            ties_in += 1   # same set, equal-score items in another order
        else:
            ties_cut += 1  # a tie at the cut: another item with the cut's score is listed
    return {"users_checked": int(got_i.shape[0]), "list_length": int(got_i.shape[1]),
            "score_rows_bit_identical": sc_same,
            "lists_identical": int(same.sum()),
            "ties_ordered_differently_inside_list": int(ties_in),
            "ties_resolved_differently_at_the_cut": int(ties_cut),
            "mismatched_users": int(bad),
            "ok": bool(sc_same and bad == 0)}
