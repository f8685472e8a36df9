#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the MI355X backend (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Workload (N = 1): BASELINE.json configs[1] -- "MovieLens-25M, als-implicit k=64, 20
epochs, 1 x MI355X".  MovieLens-25M itself is not on the box (no network), so the
input is the seeded ML-25M-shaped synthetic of ``lkpy_amd.synth``.
A *step* is ONE ALS EPOCH: user half-epoch + item half-epoch + both Gramians (+ the
exchanges when N > 1), with CSR and factors already resident in HBM.  ``value`` = epochs/s.

For N > 1 the driver launches one process per GPU (torch.distributed.run); users and
items are row-sharded (lkpy_amd._als_engine) and total work is fixed => "strong".

One JSON line on rank 0.  Extra objects (rank 0, N = 1):
  roofline     -- the dominant kernel (f32 MFMA bound): algorithmic flops per launch
                   / average launch duration from HIP events recorded on the
                  launch stream inside the library; ``frac_executed`` counts the MFMA work
                  actually issued (upper tiles only: (NT+1)/(2 NT) of the 2k^2 convention).
  parity       -- GPU vs oracle FROM IDENTICAL INPUTS at full scale: every ALS row of one
                  epoch (with the float64 referee and cond(A) accounting of
                  oracle/parity.py), sampled item-kNN rows bitwise.
  cpu_baseline -- the CPU oracle (a port of the reference's Rust + LAPACK path) timed on this
                  box's host cores: the very half-epochs of the parity leg (all rows).
  fit          -- what ``north_star`` names: ``ImplicitMFScorer(...).train(dataset)`` end to
                  end (matrix preparation, upload, relabel + transpose in HBM, plans, epochs,
                  download), with the per-epoch times of the reference's own log line.
  topk / knn   -- the other two legs of the metric, each with roofline + cpu_baseline.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

NO_ORDER_AB = False  # --no-order-ab
F32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: peak FP32 (matrix)
HBM_PEAK_GBS = 8000.0  # same guide: HBM3E peak 8 TB/s
LONG_ROW = int(os.environ.get("LK_BENCH_LONG_ROW", 2048))  # LK_ALS_LONG_ROW (csrc/als_plan.h)
CHUNK = 1024  # LK_ALS_CHUNK
# the CG leg's stopping rule ||r|| <= tol ||y||: the error of a row is up to cond(A) * tol, and
# cond(A) ~ 200 on the trained state -- 1e-6 (rounds 3-5) left the worst rows AT 1e-4 (9.6e-5 in the
# driver's run, 23 rows over in another).  2.5e-7 and 1e-7 give the SAME worst user row (1.9e-5: the
# float32 floor of the two solvers' difference) at 10.6 and 11.1 ms per epoch; the item half's
# 9.5e-5 is not CG error but that 1.9e-5 propagated -- its input P is the CG engine's own user half
CG_TOL = float(os.environ.get("LK_BENCH_CG_TOL", 2.5e-7))


WB_MAX_N = 128  # rows this short take the Woodbury kernels at padded k = 256 (csrc/als_wb.hip:
#                n <= 16, one MFMA tile; als_wb64_kernel in csrc/als_chol.hip: 17 .. 64;
#                als_wb128_kernel in csrc/als_blk.hip: 65 .. 128, unless LK_ALS_WB128=0)


def _wb_max(k: int) -> int:
    """longest row the Woodbury kernels take: 128 entries at padded k = 256; at padded k = 128
    64 (LK_ALS_WB64_K128, csrc/als_plan.h: 32 x 32 / 64 x 64 systems of als_wb64_kernel), 16
    with LK_ALS_WB64=0"""
    if k <= 128:
        if os.environ.get("LK_ALS_WB64", "1") == "0":
            return 16
        lim = int(os.environ.get("LK_ALS_WB64_K128", "64"))
        return 64 if lim >= 64 else (32 if lim >= 32 else 16)
    return WB_MAX_N if os.environ.get("LK_ALS_WB128", "1") != "0" else 64


def half_flops(lengths: np.ndarray, k: int, wb: bool = False):
    """
    Algorithmic flops of one half-epoch:
    nnz*(2k^2 + 2k) + rows_nonempty*(k^3/3 + 2k^2), split into the part done by the
    solve kernel (short rows + every solve) and by the chunk kernel (long rows' Gram).
    ``wb``: rows with <= 64 entries are solved through the Woodbury identity -- their flops are
    that method's (S0 = 2 n^2 k, S0 w and x = 4 n k, the n x n solve n^3/3 + 2 n^2), not the
    k^3/3 of a dense factorisation nobody performs (`reference_half_flops` keeps that figure).
    """
    lengths = lengths.astype(np.int64)
    per_nnz = 2 * k * k + 2 * k
    per_row = k**3 / 3.0 + 2 * k * k
    long_nnz = int(lengths[lengths > LONG_ROW].sum())
    if not wb:
        short_nnz = int(lengths.sum()) - long_nnz
        nonempty = int((lengths > 0).sum())
        return short_nnz * per_nnz + nonempty * per_row, long_nnz * per_nnz
    dense = lengths[lengths > _wb_max(k)]
    n = lengths[(lengths > 0) & (lengths <= _wb_max(k))].astype(np.float64)
    wb_flops = float((2 * n * n * k + 4 * n * k + n**3 / 3.0 + 2 * n * n).sum())
    return (int(dense.sum()) - long_nnz) * per_nnz + len(dense) * per_row + wb_flops, \
        long_nnz * per_nnz


def reference_half_flops(lengths: np.ndarray, k: int) -> float:
    "what the reference's algorithm (a dense sposv per row) costs for this half (SURVEY 8d)"
    a, b = half_flops(lengths, k, wb=False)
    return a + b


def half_mfma_flops(lengths: np.ndarray, kp: int, wb: bool = False):
    """
    Matrix-core flops the solve kernel actually ISSUES in one half-epoch: v_mfma_f32_16x16x4
    instructions x 2048.  Gram: one instruction per upper tile per group of 4 entries
    (NT(NT+1)/2 per group; long rows are the chunk kernel's); for padded k > 64 also the
    blocked Cholesky's trailing updates, 4 * sum_b (NT-1-b)(NT-b)/2 per row
    (csrc/als_blk.hip).  The k <= 64 kernel factorises in 4-column panels (VALU) with MFMA
    trailing updates (79 instructions per row at k = 64, full tiles): counted as the k^3/3
    useful flops per row, not as what the matrix cores execute for it.  ``wb``: a Woodbury row
    issues kp/4 MFMAs per 16 x 16 tile of S0 (plus the 64 x 64 solve's for 17 .. 64 entries).
    """
    lengths = lengths.astype(np.int64)
    nt = kp // 16
    tri = nt * (nt + 1) // 2
    wb_mfma = 0
    if wb:
        # n <= 16: one S0 tile over all features (kp/4 instructions); 17 .. 64: ceil(n/16)^2
        # tiles + the 79 trailing-update instructions of the 64 x 64 hybrid solve
        n16 = int(((lengths > 0) & (lengths <= 16)).sum())
        mid = lengths[(lengths > 16) & (lengths <= min(64, _wb_max(kp)))]
        big = lengths[(lengths > 64) & (lengths <= _wb_max(kp))]
        # 65 .. 128: all 36 upper tiles of the 128 x 128 system over kp / 4 steps + the k = 128
        # blocked solver's trailing updates (4 * 84 instructions)
        wb_mfma = n16 * (kp // 4) + int((((mid + 15) // 16) ** 2).sum()) * (kp // 4) \
            + 79 * len(mid) + len(big) * (36 * (kp // 4) + 4 * 84)
        lengths = lengths[lengths > _wb_max(kp)]
    short = lengths[(lengths > 0) & (lengths <= LONG_ROW)]
    groups = int(((short + 3) // 4).sum())
    nonempty = int((lengths > 0).sum())
    fl = groups * tri * 2048.0 + wb_mfma * 2048.0
    if kp > 64:
        chol = 4 * sum((nt - 1 - b) * (nt - b) // 2 for b in range(nt))
        fl += nonempty * chol * 2048.0
    else:
        fl += nonempty * (kp**3 / 3.0)
    return fl


def half_bytes(lengths: np.ndarray, k: int):
    "Algorithmic HBM bytes of one half-epoch (SURVEY.md section 8d)."
    nnz, rows = int(lengths.sum()), len(lengths)
    return nnz * (4 + 4 + 4 * k) + (rows + 1) * 4 + rows * k * 4 * 2 + k * k * 4


def pmc_traffic(pattern: str, kernel_substr: str):
    """
    HBM bytes per launch of a kernel from the newest COMMITTED rocprofv3 PMC summary
    (profiles/<pattern>, written by tools/prof_*.sh + tools/summarize_prof.py on the same
    workload): (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- FETCH_SIZE is doubled because on gfx950
    it reports half the bytes of wide (16 B/lane) coalesced reads (MI355X_MICROARCH.md,
    section HBM).  Counters cannot be collected inside this process, so the value is null
    when no summary is committed.
    """
    import csv

    pats = [pattern] if isinstance(pattern, str) else list(pattern)
    files = sorted({f for pat in pats for f in (ROOT / "profiles").glob(pat)},
                   key=lambda f: f.name)
    if not files:
        return None, None
    rows = []
    with open(files[-1]) as f:
        for row in csv.DictReader(f):
            if kernel_substr in row["Kernel_Name"] and row["Counter_Name"] in ("FETCH_SIZE",
                                                                              "WRITE_SIZE"):
                rows.append((float(row.get("Grid_Size") or 0), row["Counter_Name"],
                             float(row["mean"])))
    if not rows:
        return None, None
    # the launches of the kernel proper: since round 5 a half-epoch also has a SMALL launch of the
    # same kernel for the long rows (a few hundred workgroups: their right-hand sides come from
    # the chain kernel) -- averaging it in would halve the figure
    gmax = max(r[0] for r in rows)
    fetch = [m for g, c, m in rows if c == "FETCH_SIZE" and g >= 0.05 * gmax]
    write = [m for g, c, m in rows if c == "WRITE_SIZE" and g >= 0.05 * gmax]
    if not fetch:
        return None, None
    w = sum(write) / len(write) if write else 0.0
    return (2.0 * sum(fetch) / len(fetch) + w) * 1024.0, files[-1].name


def pmc_traffic_per_epoch(pattern, anchor_kernel: str):
    """
    HBM bytes per EPOCH summed over every kernel of a committed PMC summary (a workload whose
    epoch is several kernels: cfg5's dense + chunk + Woodbury + Z kernels): sum over kernels of
    (2 * FETCH_SIZE + WRITE_SIZE) * 1024 * launches, divided by the epochs in the capture
    (= launches of ``anchor_kernel`` / 2: one per half-epoch).
    """
    import csv

    pats = [pattern] if isinstance(pattern, str) else list(pattern)
    files = sorted({f for pat in pats for f in (ROOT / "profiles").glob(pat)},
                   key=lambda f: f.name)
    if not files:
        return None, None
    tot, anchor = 0.0, 0
    with open(files[-1]) as f:
        for row in csv.DictReader(f):
            n = float(row.get("count", 0) or 0)
            if row["Counter_Name"] == "FETCH_SIZE":
                tot += 2.0 * float(row["mean"]) * n * 1024.0
                if anchor_kernel in row["Kernel_Name"]:
                    anchor += int(n)
            elif row["Counter_Name"] == "WRITE_SIZE":
                tot += float(row["mean"]) * n * 1024.0
    if anchor < 2 or tot <= 0:
        return None, None
    return tot / (anchor / 2.0), files[-1].name


TOPK_KERNELS = ("score_filter", "sample_cmax", "cmax_tau", "cand_select", "sample_rows", "sample_tau",
                "score_panel", "score_mask", "row_topn")


def pmc_traffic_topk(pattern="r*topk_counters.csv"):
    """
    HBM bytes per ``lk_score_topk`` call from the newest committed PMC summary of the top-N leg:
    the sum over its kernels of (2 * FETCH_SIZE + WRITE_SIZE) * 1024 * launches, divided by the
    calls in the capture (= launches of ``sample_rows_kernel``, one per fused call).
    """
    import csv

    files = sorted((ROOT / "profiles").glob(pattern), key=lambda f: f.name)
    if not files:
        return None, None
    tot, calls = 0.0, 0
    with open(files[-1]) as f:
        for row in csv.DictReader(f):
            if not any(kn in row["Kernel_Name"] for kn in TOPK_KERNELS):
                continue
            n = float(row.get("count", 0) or 0)
            if row["Counter_Name"] == "FETCH_SIZE":
                tot += 2.0 * float(row["mean"]) * n * 1024.0
                if "sample_rows_kernel" in row["Kernel_Name"]:
                    calls += int(n)
            elif row["Counter_Name"] == "WRITE_SIZE":
                tot += float(row["mean"]) * n * 1024.0
    if calls < 1 or tot <= 0:
        return None, None
    return tot / calls, files[-1].name


def reference_order_recheck(sub, other, otor, want_rows, k, dev):
    """
    REPRODUCE exception rows instead of refereeing them: the listed rows once more through a
    REFERENCE-ORDER plan (``lk_als_plan_create_ex`` + rhs workspace: y as one sequential float32
    chain per feature, the normal matrix in matrixmultiply's 256-entry blocks added in order --
    src/accel/als/implicit.rs:110-117) from the very inputs the oracle had -- ``sub``: the rows'
    CSR with the entries in the oracle's order, ``other``: the gathered factor matrix (host array
    or padded device tensor, same labelling as ``sub``'s columns), ``otor``: the oracle's OtOr --
    and compared with the oracle's rows.
    """
    import torch

    from lkpy_amd import _device as D
    from lkpy_amd import _native

    csr = D.DeviceCSR.from_arrays(sub.indptr.astype(np.int64), sub.indices.astype(np.int32),
                                  sub.data.astype(np.float32), sub.shape, dev)
    plan = D.ALSPlan(csr, k, _native.SOLVER_CHOLESKY, reference_order=True)
    d_other = other if isinstance(other, torch.Tensor) else D.to_device_padded(other, dev)
    d_otor = torch.from_numpy(np.ascontiguousarray(otor, dtype=np.float32)).to(dev)
    this = torch.zeros((sub.shape[0], plan.kp), dtype=torch.float32, device=dev)
    plan.half_epoch(this, d_other, d_otor)
    plan.check_status()
    got = D.to_host_unpadded(this, k)
    num = np.linalg.norm(got.astype(np.float64) - want_rows, axis=1)
    den = np.linalg.norm(want_rows.astype(np.float64), axis=1)
    rel = num / np.maximum(den, 1e-300)
    return {"rows": int(len(rel)), "within_1e-4": int((rel <= 1.0e-4).sum()),
            "rel_max": float(rel.max()) if len(rel) else 0.0,
            "rel": [float(x) for x in rel[:10]]}


def als_parity_and_cpu(eng, ui, k, reg, row_frac: float):
    """
    One more epoch on the GPU from the trained state, and the SAME two half-epochs on the CPU
    oracle from identical inputs (user half from (P, Q); item half from (Q, P_new as the GPU
    produced it)), timed: the oracle run is both the parity reference and the cpu_baseline.
    ``row_frac`` < 1 (large k: the oracle's cost grows with k^2..k^3) restricts both sides to
    a seeded row sample; the kernel still ran every row.
    """
    import scipy.sparse as sps

    from oracle import lk_oracle as lko
    from oracle import parity

    P, Q = eng.user_embeddings(), eng.item_embeddings()
    eng.train_epoch()
    eng.check()
    P1, Q1 = eng.user_embeddings(), eng.item_embeddings()
    iu = sps.csr_array(ui.T)
    iu.sort_indices()
    # how many threads?  The port's row-parallel loop calls SciPy's bundled OpenBLAS sposv from
    # every thread; on the 256-thread bench host MORE threads were SLOWER (tools/oracle_threads.py,
    # 5 M entries: 0.18 s at 8 threads, 0.33 s at 32, 0.81 s at 64, a crash at 128) -- and a short
    # sample does not show it reliably (30 ms samples picked 8, 16 and 32 in three runs of this
    # file while the full user half took 1.0 / 1.1 / 2.0 s).  So the USER HALF ITSELF is timed at
    # 8 / 16 / 32 threads when every row is checked (row results do not depend on the thread
    # count), the fastest count is the baseline's, and all three times are reported.
    cap = lko.num_threads()
    cands = sorted({min(8, cap), min(16, cap), cap})
    tried = {}
    if row_frac >= 1.0:
        otor_u = lko.implicit_otor(Q, reg)
        for t_ in cands:
            w_ = np.ascontiguousarray(P.copy())
            t0 = time.perf_counter()
            lko.als_half_epoch(ui, w_, Q, otor_u, t_)
            tried[t_] = time.perf_counter() - t0
        threads = min(tried, key=tried.get)
    else:
        threads = min(8, cap)  # (sampled legs: the reference's own default, min(ncpus, 8))
    rng = np.random.default_rng(5)
    out, secs, desc = {}, 0.0, []
    for name, mat, this, other, got in (("user", ui, P, Q, P1), ("item", iu, Q, P1, Q1)):
        n = mat.shape[0]
        if row_frac < 1.0:
            rows = np.sort(rng.choice(n, max(256, int(n * row_frac)), replace=False))
            sub, t0_, got_ = sps.csr_array(mat[rows]), this[rows], got[rows]
        else:
            rows, sub, t0_, got_ = None, mat, this, got
        want = np.ascontiguousarray(t0_.copy())
        otor = lko.implicit_otor(other, reg)
        t0 = time.perf_counter()
        lko.als_half_epoch(sub, want, other, otor, threads)
        dt = time.perf_counter() - t0
        full = dt * mat.nnz / max(sub.nnz, 1)
        secs += full
        desc.append(f"{name} half: {sub.shape[0]} of {n} rows ({sub.nnz} nnz) in {dt:.2f}s")
        exact, cond = lko.als_referee_f64(sub, other, reg)
        acc = parity.als_half_accounting(got_, want, exact, cond)
        acc.pop("by_cond_decade", None)
        if acc.get("exceptions"):
            # the rows over 1e-4, once more through a reference-order plan: same rows, same entry
            # order (the oracle's, i.e. the ORIGINAL labelling -- the engine relabels rows, and
            # the order of a row's entries is part of the reference's arithmetic), same OtOr
            ex = np.array([e["row"] for e in acc["exceptions"]])
            for e in acc["exceptions"]:
                e["entries"] = int(sub.indptr[e["row"] + 1] - sub.indptr[e["row"]])
            acc["exceptions_in_reference_order"] = reference_order_recheck(
                sps.csr_array(sub[ex]), other, otor, want[ex], k, eng.backend.dev)
        out[name] = acc
    exceptions = [dict(e, half=name) for name, o in out.items() for e in o.get("exceptions", [])]
    reco = [o["exceptions_in_reference_order"] for o in out.values()
            if "exceptions_in_reference_order" in o]
    par = {
        "what": "one epoch from the trained state, GPU vs oracle from identical inputs, "
        + ("every row" if row_frac >= 1.0 else f"a {row_frac:.3f} row sample"),
        "criterion": "ok = no row further than 1e-4 (relative) from the oracle's row -- the raw "
        "north-star tolerance, nothing folded in; rows over it are listed in `exceptions` with "
        "cond(A) and their distances to the float64 referee; `accounted` is the separate "
        "statement that each exception is the reference arithmetic's own deviation",
        "als_rel_P": out["user"]["rel_gpu_vs_oracle"],
        "als_rel_Q": out["item"]["rel_gpu_vs_oracle"],
        "rows_checked": out["user"]["rows"] + out["item"]["rows"],
        "rows_over_1e-4": out["user"]["rows_over_1e-4"] + out["item"]["rows_over_1e-4"],
        "row_rel_max": max(out["user"]["row_rel_max"], out["item"]["row_rel_max"]),
        "ok": bool(all(o["ok"] for o in out.values())),
        "exceptions": exceptions,
        "ok_in_reference_order": bool(all(r["within_1e-4"] == r["rows"] for r in reco)),
        "exceptions_reproduced_in_reference_order": [sum(r["within_1e-4"] for r in reco),
                                                     sum(r["rows"] for r in reco)],
        "accounted": bool(all(o["accounted"] for o in out.values())),
        "rows_decidable": sum(o.get("rows_decidable", 0) for o in out.values()),
        "gpu_row_err_over_cond_u_max": max(o["row_err_over_cond_u_max_gpu"] for o in out.values()),
        "oracle_row_err_over_cond_u_max": max(
            o["row_err_over_cond_u_max_oracle"] for o in out.values()),
        "user": out["user"],
        "item": out["item"],
    }
    cpu = {
        "value": 1.0 / secs,
        "unit": "epochs/s",
        "cores": threads,
        "kind": "port",
        "sample": "; ".join(desc) + ("" if row_frac >= 1.0 else "; extrapolated by nnz"),
        "host_cpus": os.cpu_count(),
        "cpu_seconds_per_epoch": round(secs, 3),
    }
    cpu["threads_tried_user_half_seconds"] = {str(t_): round(v, 3) for t_, v in tried.items()}
    cpu["threads_note"] = (
        "`cores` = the fastest of 8 / 16 / %d threads for the FULL user half (the port's "
        "row-parallel loop calls SciPy's bundled OpenBLAS sposv from every thread; beyond ~8-16 "
        "callers it slows down and at 128 it crashes -- tools/oracle_threads.py); sampled legs "
        "use min(ncpus, 8), the reference's own default (src/lenskit/schemas/settings.py:182-185); "
        "BASELINE.md section 2 names $(nproc) = %d here" % (cap, os.cpu_count() or 0))
    return par, cpu


def fit_leg(ratings, k, epochs, weight, keep: dict | None = None):
    """
    ``ImplicitMFScorer(embedding_size=k, epochs=epochs).train(dataset)``: the call
    ``north_star`` names, end to end, and the per-epoch wall times the reference logs
    (src/lenskit/training.py:320-329, ``finished epoch ... time=``).
    """
    import logging

    import torch

    from lkpy_amd import training
    from lkpy_amd.als import ImplicitMFScorer
    from lkpy_amd.data import Dataset, Vocabulary
    from lkpy_amd.training import TrainingOptions

    n_users, n_items = ratings.shape
    rows = np.repeat(np.arange(n_users, dtype=np.int32), np.diff(ratings.indptr))
    ds = Dataset(Vocabulary(np.arange(n_users), "user", reorder=False),
                 Vocabulary(np.arange(n_items), "item", reorder=False),
                 rows, ratings.indices, {"rating": ratings.data})

    class Tap(logging.Handler):
        times: list = []

        def emit(self, record):
            if "finished epoch" in record.getMessage() and len(record.args) >= 2:
                self.times.append(float(record.args[1]))

    tap = Tap()
    tap.times = []
    training._log.addHandler(tap)
    old_level = training._log.level
    training._log.setLevel(logging.INFO)
    try:
        scorer = ImplicitMFScorer(embedding_size=k, epochs=epochs, weight=weight)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        scorer.train(ds, TrainingOptions(rng=42))
        torch.cuda.synchronize()
        fit = time.perf_counter() - t0
    finally:
        training._log.removeHandler(tap)
        training._log.setLevel(old_level)
    assert scorer.item_embeddings.shape == (n_items, k)
    if keep is not None:
        keep["scorer"], keep["ds"] = scorer, ds  # the recommend leg serves from this model
    ep = tap.times
    return {
        "what": f"ImplicitMFScorer(embedding_size={k}, epochs={epochs}).train(dataset), host "
        "Dataset in, host factor arrays out",
        "fit_seconds": round(fit, 4),
        "epoch_seconds_logged": [round(t, 5) for t in ep],
        "epochs_per_s_from_log": round(len(ep[1:]) / sum(ep[1:]), 2) if len(ep) > 1 else None,
        "setup_and_download_seconds": round(fit - sum(ep), 4) if ep else None,
    }



def recommend_leg(scorer, ds, n_users=10000, n=100, no_cpu=False):
    """
    ``pipelines/als-implicit.toml``'s recommend path for a BATCH of users at the cfg2 scale, as
    the reference runs it per query (src/lenskit/batch/_runner.py:283-308): training history
    (basic/history.py:77-95) -> the user re-solved from that history (``new_user_embedding``,
    als/_implicit.py:77-130, ``_OtOr`` from the trained state) -> scores of all items
    (als/_common.py:159-170) -> candidates minus the history (basic/candidates.py:77-94) ->
    top-``n`` (accel/data/sorting.rs:132-172).  Here: ``UserTrainingHistoryLookup.batch`` +
    ``ImplicitMFScorer.recommend_batch`` -- one row-gather launch, one fold-in half-epoch, one
    fused score + top-N call.  The model is the one the ``fit`` leg trained through the component.
    """
    import torch

    from lkpy_amd import _device as D
    from lkpy_amd import batch as lk_batch
    from lkpy_amd.basic import UserTrainingHistoryLookup
    from lkpy_amd.pipeline import Pipeline

    pipe = Pipeline.load_config(ROOT / "tests" / "golden" / "pipelines" / "als-implicit.toml")
    pipe.node("scorer").component = scorer           # the trained model in the TOML's pipeline
    lookup = pipe.node("history-lookup").component
    lookup.train(ds)
    pipe.node("candidate-selector").component.train(ds)
    rng = np.random.default_rng(11)
    users = rng.choice(ds.users.ids(), min(n_users, ds.user_count), replace=False)
    k = scorer.config.embedding_size
    dev = D.device()
    sync = lambda: torch.cuda.synchronize(dev)  # noqa: E731

    scorer.recommend_batch(lookup.batch(users[:512]), n)  # uploads: training matrix, Q, OtOr
    walls, walls_ilc = [], []
    for _ in range(5):
        sync()
        t0 = time.perf_counter()
        hb = lookup.batch(users)
        g_idx, g_sc = scorer.recommend_batch(hb, n)
        walls.append(time.perf_counter() - t0)
    for _ in range(3):
        sync()
        t0 = time.perf_counter()
        ilc = lk_batch.recommend(pipe, users, n)
        walls_ilc.append(time.perf_counter() - t0)
    wall = min(walls)
    assert len(ilc) == len(users) and len(ilc.lookup(users[0].item())) == n

    # the device work of the same call, piece by piece, on the launch stream (torch's current
    # stream IS the stream every lk_* call is given): HIP events around each piece, a spin kernel
    # ahead of it so the host is done enqueuing before the first kernel of the piece starts
    st = scorer._device_state()
    hb = lookup.batch(users)
    lens = hb.lengths

    def timed(fn, reps=5):
        best = None
        for _ in range(reps):
            sync()
            torch.cuda._sleep(4_000_000)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            res = fn()
            b.record()
            sync()
            ms = a.elapsed_time(b)
            best = ms if best is None or ms < best else best
        return best, res

    src = lookup._device_matrix()["csr"]
    if not scorer.config.use_ratings:  # implicit: the constant weight, no rating read
        src = D.DeviceCSR(src.indptr, src.indices, None, src.shape, src.h_indptr)
    t_gather, hist = timed(lambda: D.gather_rows(src, hb.user_nums, scale=scorer.config.weight))
    plan = D.ALSPlan(hist, k, scorer._solver(), reference_order="accurate")  # (as D.fold_in does)
    u = torch.zeros((len(users), plan.kp), dtype=torch.float32, device=dev)
    t_fold, _ = timed(lambda: plan.half_epoch(u, st["Q"], st["OtOr"]))
    plan.check_status()
    t_topk, (d_idx, d_sc) = timed(lambda: D.score_topk(u, st["Q"], k, n, hist.indptr,
                                                        hist.indices))
    gpu_ms = t_gather + t_fold + t_topk
    t_down0 = time.perf_counter()
    D.to_host(torch.cat([d_idx.view(torch.float32), d_sc], dim=1))
    t_down = time.perf_counter() - t_down0
    fold_flops = sum(half_flops(lens, k))
    topk_flops = 2.0 * len(users) * len(ds.items) * k
    res = {
        "metric": "als-implicit.toml batch recommend: %d users' training histories -> batched "
                  "fold-in -> scores of all %d items -> top-%d without the history, seconds"
                  % (len(users), len(ds.items), n),
        "value": round(wall, 5), "unit": "s", "higher_is_better": False,
        "users": int(len(users)), "users_per_s": round(len(users) / wall, 1),
        "history_entries": int(lens.sum()),
        "call": "scorer.recommend_batch(lookup.batch(users), n): user ids in, host [B x n] item "
                "numbers + scores out (download included)",
        "seconds_through_batch_recommend": round(min(walls_ilc), 5),
        "device_ms": {"gather_histories": round(t_gather, 4), "fold_in": round(t_fold, 4),
                      "score_topn": round(t_topk, 4), "sum": round(gpu_ms, 4)},
        "download_seconds": round(t_down, 5),
        "host_seconds": round(max(wall - gpu_ms / 1e3 - t_down, 0.0), 5),
        "host_fraction_of_call": round(max(wall - gpu_ms / 1e3 - t_down, 0.0) / wall, 4),
        "roofline": {
            "kernel": "the fold-in launch (lk_als_implicit_half_epoch over the histories' CSR)",
            "bound": "mfma", "achieved": round(fold_flops / (t_fold * 1e-3) / 1e12, 3),
            "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(fold_flops / (t_fold * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
            "algorithmic_flops": fold_flops, "traffic": None,
            "score_topn_frac": round(topk_flops / (t_topk * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
            "note": "nnz (2k^2 + 2k) + rows (k^3/3 + 2k^2) over the batch's histories / HIP-event "
                    "time of the launch; 10^4 rows fill the chip for a fraction of a millisecond, "
                    "so launch ramp and tail are a visible part of it",
        },
    }
    # HBM bytes per fold-in launch from the committed PMC summary of the same call
    # (tools/prof_recommend.sh; the fold-in plan's offsets are 64-bit: the <4, true, ...> instance)
    res["roofline"]["traffic"], res["roofline"]["traffic_source"] = pmc_traffic(
        "r*_recommend_counters.csv", "als_solve_kernel<4, true")
    res["roofline"]["algorithmic_bytes"] = half_bytes(lens, k)
    if no_cpu:
        return res
    from oracle import lk_oracle as lko
    from oracle import parity

    # the reference's per-query path on the host, from the same trained model
    Q = np.ascontiguousarray(scorer.item_embeddings)
    OtOr = np.ascontiguousarray(scorer._OtOr)
    hp = ds._indptr
    w32 = np.float32(scorer.config.weight)
    u_gpu = D.to_host_unpadded(u, k)
    nums = hb.user_nums
    want_u = np.zeros_like(u_gpu)
    t0 = time.perf_counter()
    for r, un in enumerate(nums):
        items = ds._cols[hp[un]:hp[un + 1]]
        want_u[r] = lko.als_fold_in(items, np.full(len(items), w32, np.float32), Q, OtOr)
    t_fold_cpu = time.perf_counter() - t0
    num = np.linalg.norm(u_gpu.astype(np.float64) - want_u, axis=1)
    den = np.linalg.norm(want_u.astype(np.float64), axis=1)
    rel = num / np.maximum(den, 1e-300)
    # lists: the oracle's score + exclusion + heap top-n FROM THE GPU'S query vectors
    threads = lko.num_threads()
    m = min(len(users), 4096)
    ptr = np.zeros(m + 1, np.int64)
    np.cumsum(lens[:m], out=ptr[1:])
    ex = np.concatenate([ds._cols[hp[un]:hp[un + 1]] for un in nums[:m]]).astype(np.int32)
    lko.score_topn_batch(Q, u_gpu[:threads], n, None, None, threads)  # page-in
    t0 = time.perf_counter()
    want_i, want_s = lko.score_topn_batch(Q, u_gpu[:m], n, ptr, ex, threads)
    t_list_cpu = time.perf_counter() - t0
    lists = parity.topn_accounting(np.asarray(g_idx)[:m], np.asarray(g_sc)[:m], want_i, want_s,
                                   lambda r: lko.score_dense(Q, u_gpu[r]))
    # one thread, the reference's loop: fold-in then score + top-n per query
    m1 = min(m, 512)
    t0 = time.perf_counter()
    lko.score_topn_batch(Q, u_gpu[:m1], n, ptr[:m1 + 1], ex[:ptr[m1]], 1)
    t_list_1 = (time.perf_counter() - t0) / m1
    per_user_1 = t_fold_cpu / len(users) + t_list_1
    res["cpu_baseline"] = {
        "value": round(per_user_1 * len(users), 3), "unit": "s", "cores": 1, "kind": "port",
        "sample": f"the reference's per-query loop on ONE thread -- its batch runner's default on a "
                  f"GIL interpreter: num_batch_jobs = 1 (schemas/settings.py:190-194) -> "
                  f"_sequential_results (batch/_runner.py:281-288): fold-in of all {len(users)} "
                  f"users ({t_fold_cpu:.2f}s, NumPy + SciPy "
                  f"cho_factor as _implicit.py:101-130) + score / exclusion / heap top-{n} of "
                  f"{m1} users ({t_list_1 * m1:.2f}s), extrapolated by users",
        "score_topn_all_threads_seconds": round(t_list_cpu / m * len(users), 3),
        "threads": threads,
    }
    res["parity"] = {
        "fold_in": {"rows_checked": int(len(rel)), "rows_over_1e-4": int((rel > 1e-4).sum()),
                    "row_rel_max": float(rel.max()), "row_rel_p50": float(np.median(rel)),
                    "ok": bool((rel <= 1e-4).all()),
                    "criterion": "every fold-in vector within 1e-4 (relative, raw) of the "
                                 "oracle's _train_new_row restatement from the same model"},
        "lists": lists,
        "ok": bool((rel <= 1e-4).all() and lists["ok"]),
    }
    return res


def topk_cpu_and_parity(P, Q, ex_ptr, ex_idx, gpu_idx, gpu_sc, n, budget_s=10.0,
                        max_users=32768, block=1024):
    """
    bench.py's checker leg for the dense top-N call.  The reference's path for one user --
    scores = Q @ u (``ALSBase.__call__``), candidates = all items minus the user's own, heap
    top-N -- through the oracle's restatement (``lko_score_topn_batch``: the per-query loop on
    all host threads, as the reference's batch runner spreads queries over workers), on a seeded
    user sample: TIMED (cpu_baseline, extrapolated by users) and COMPARED with the lists the GPU
    produced for the same users from the same factors (parity: index lists and score bits).
    ``P``/``Q``/exclusions are in the engine's (relabelled) row order on both sides.
    """
    from oracle import lk_oracle as lko

    rng = np.random.default_rng(7)
    n_users = P.shape[0]
    users = rng.choice(n_users, min(n_users, max_users), replace=False)
    threads = lko.num_threads()
    ex_ptr = np.asarray(ex_ptr, dtype=np.int64)

    def run(us):
        lens = ex_ptr[us + 1] - ex_ptr[us]
        ptr = np.zeros(len(us) + 1, np.int64)
        np.cumsum(lens, out=ptr[1:])
        idx = np.concatenate([ex_idx[ex_ptr[u]:ex_ptr[u + 1]] for u in us]) if len(us) else \
            np.empty(0, np.int32)
        return lko.score_topn_batch(Q, P[us], n, ptr, idx, threads)

    run(users[:threads])  # first call: library load, page-in
    done, blocks = 0, []
    t0 = time.perf_counter()
    while done < len(users):
        us = users[done:done + block]
        blocks.append(run(us))
        done += len(us)
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    users = users[:done]
    want_i = np.concatenate([b[0] for b in blocks])
    want_s = np.concatenate([b[1] for b in blocks])
    cpu = {"value": round(dt / done * n_users, 2), "unit": "s", "cores": threads, "kind": "port",
           "sample": f"{done} of {n_users} users in {dt:.2f}s (score_dense + exclusion + heap "
           f"top-N per user, the reference's per-query path, queries spread over {threads} "
           "threads), extrapolated by users"}
    from oracle import parity

    par = parity.topn_accounting(gpu_idx[users], gpu_sc[users], want_i, want_s,
                                 lambda r: lko.score_dense(Q, P[users[r]]))
    return cpu, par



def knn_component_recommend(ratings, gpu_lists=None, n_users=10000, n=100):
    """
    ``pipelines/iknn-explicit.toml``'s recommend path for a batch of users THROUGH THE COMPONENTS
    (the ``knn.recommend`` leg above calls ``lk_iknn_recommend`` on arrays the bench prepared):
    ``Pipeline.load_config`` -> ``train`` (``save_nbrs = 100``, ``max_nbrs = 100``: the model of
    that leg) -> ``batch.recommend(pipe, users, 100)`` = ``UserTrainingHistoryLookup.batch`` (user
    numbers) + ``ItemKNNScorer.recommend_batch`` (histories gathered and mean-centred on the
    device, hit counts per user from the device).  Same users as the array-level leg.
    """
    import torch

    from lkpy_amd import batch as lk_batch
    from lkpy_amd.data import Dataset, Vocabulary
    from lkpy_amd.pipeline import Pipeline

    n_u, n_i = ratings.shape
    rows = np.repeat(np.arange(n_u, dtype=np.int32), np.diff(ratings.indptr))
    ds = Dataset(Vocabulary(np.arange(n_u), "user", reorder=False),
                 Vocabulary(np.arange(n_i), "item", reorder=False),
                 rows, ratings.indices, {"rating": ratings.data})
    pipe = Pipeline.load_config(ROOT / "tests" / "golden" / "pipelines" / "iknn-explicit.toml")
    scorer = pipe.node("scorer").component
    scorer.config.save_nbrs, scorer.config.max_nbrs, scorer.config.min_nbrs = 100, 100, 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.train(ds)
    torch.cuda.synchronize()
    t_train = time.perf_counter() - t0
    users = np.random.default_rng(43).choice(n_u, min(n_users, n_u), replace=False)
    lk_batch.recommend(pipe, users[:256], n)  # uploads: training matrix, item means, hit counts
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = lk_batch.recommend(pipe, users, n)
        ts.append(time.perf_counter() - t0)
    res = {"what": "iknn-explicit.toml (save_nbrs = max_nbrs = 100) trained through the pipeline, "
                   "batch.recommend(pipe, 10 000 user ids, 100): ids in, array-backed "
                   "ItemListCollection out",
           "seconds": round(min(ts), 5), "users": int(len(users)),
           "users_per_s": round(len(users) / min(ts), 1),
           "pipeline_train_seconds": round(t_train, 3), "listed": int(out.total_items())}
    return res

def _gather_rows(out, rows):
    "rows of a DeviceCSR similarity matrix -> host (ptr, idx, val); torch as plumbing only"
    import torch

    r = torch.as_tensor(rows.astype(np.int64), device=out.indptr.device)
    beg, end = out.indptr[r], out.indptr[r + 1]
    lens = end - beg
    ptr = torch.zeros(len(rows) + 1, dtype=torch.int64, device=lens.device)
    ptr[1:] = torch.cumsum(lens, 0)
    pos = torch.arange(int(ptr[-1].item()), device=lens.device)
    src = pos - torch.repeat_interleave(ptr[:-1], lens) + torch.repeat_interleave(beg, lens)
    return ptr.cpu().numpy(), out.indices[src].cpu().numpy(), out.values[src].cpu().numpy()


def knn_cpu_and_parity(dui, diu, out):
    """
    bench.py's checker leg (the ONLY place this module touches ``oracle/``): the oracle's
    ``sim_row`` (port of src/accel/knn/item_train.rs:95-152) on a seeded row sample of the SAME
    normalised matrices -- timed (cpu_baseline, extrapolated by multiply-accumulates) and
    compared bitwise with the GPU's rows (parity).
    """
    import scipy.sparse as sps
    import torch  # noqa: F401

    from oracle import lk_oracle as lko
    from oracle import parity

    ui = sps.csr_array((dui.values.cpu().numpy(), dui.indices.cpu().numpy(), dui.h_indptr),
                       shape=dui.shape)
    iu = sps.csr_array((diu.values.cpu().numpy(), diu.indices.cpu().numpy(), diu.h_indptr),
                       shape=diu.shape)
    ulen = np.diff(ui.indptr).astype(np.int64)
    row_macs = np.add.reduceat(ulen[iu.indices], np.minimum(iu.indptr[:-1], iu.nnz - 1)
                               .astype(np.int64))
    row_macs[np.diff(iu.indptr) == 0] = 0
    total = int(row_macs.sum())
    rng = np.random.default_rng(2)
    threads = min(lko.num_threads(), os.cpu_count() or 1)
    rows = np.sort(rng.choice(ui.shape[1], min(2048, ui.shape[1]), replace=False)).astype(np.int32)
    t0 = time.perf_counter()
    want = lko.iknn_build_rows(ui, iu, rows, 1.0e-6, None, threads)
    dt = time.perf_counter() - t0
    frac = float(row_macs[rows].sum()) / max(total, 1)
    cpu = {
        "value": round(dt / max(frac, 1e-12), 2),
        "unit": "s",
        "cores": threads,
        "kind": "port",
        "sample": f"{len(rows)} of {ui.shape[1]} item rows ({frac * 100:.2f}% of the "
        f"multiply-accumulates) in {dt:.2f}s, extrapolated by MACs",
    }
    par = parity.knn_rows_equal(*_gather_rows(out, rows), want)
    return cpu, {"knn_rows_checked": par["rows_checked"],
                 "knn_entries_checked": par["entries_checked"],
                 "knn_bitwise_equal": par["bitwise_equal"]}



def knn_score_cpu_and_parity(sims, r_ptr, r_idx, r_val, t_ptr, t_idx, got_s, got_c):
    """
    The cfg3 batch-scoring call (10 000 queries x 100 targets, ``max_nbrs`` 100, ``min_nbrs`` 1)
    through the oracle's ``score_explicit`` restatement (src/accel/knn/item_score.rs:23-111,
    accum.rs) for EVERY query, timed, and compared with the GPU's scores / neighbour counts.
    Counts and the null pattern must be identical; scores within 1e-5 relative -- and since the
    candidate-list kernel follows the reference's accumulator step for step (same evictions among
    equal similarities, same summation order, unfused multiply-add), bit for bit:
    ``scores_bit_identical`` of ``scores_compared``.
    """
    import scipy.sparse as sps

    from oracle import lk_oracle as lko

    h = sps.csr_array((sims.values.cpu().numpy(), sims.indices.cpu().numpy(),
                       sims.indptr.cpu().numpy()), shape=sims.shape)
    threads = lko.num_threads()
    t0 = time.perf_counter()
    want_s, want_c = lko.iknn_score_batch(h, r_ptr, r_idx, r_val, t_ptr, t_idx, 100, 1, threads)
    dt = time.perf_counter() - t0
    nan_same = bool(np.array_equal(np.isnan(got_s), np.isnan(want_s)))
    fin = ~np.isnan(want_s) & ~np.isnan(got_s)
    rel = np.abs(got_s[fin].astype(np.float64) - want_s[fin]) / np.maximum(np.abs(want_s[fin]), 1e-6)
    return {
        "cpu_baseline": {"value": round(dt, 3), "unit": "s", "cores": threads, "kind": "port",
                         "sample": f"all {len(r_ptr) - 1} queries (score_explicit per query, "
                         f"queries spread over {threads} threads)"},
        "parity": {"queries_checked": int(len(r_ptr) - 1), "targets_checked": int(len(t_idx)),
                   "counts_identical": bool(np.array_equal(got_c, want_c)),
                   "null_pattern_identical": nan_same,
                   "score_rel_max": float(rel.max()) if len(rel) else 0.0,
                   "scores_over_1e-5": int((rel > 1e-5).sum()),
                   "scores_compared": int(fin.sum()),
                   "scores_bit_identical": int(np.sum(
                       got_s[fin].view(np.uint32) == want_s[fin].astype(np.float32).view(np.uint32))),
                   "ok": bool(nan_same and np.array_equal(got_c, want_c)
                              and (len(rel) == 0 or rel.max() <= 1e-5))},
    }


def knn_recommend_cpu_and_parity(sims, r_ptr, r_idx, r_val, means, hits, got_i, got_s, n,
                                 budget_s=10.0):
    """
    The cfg3 recommend call (top-100 for 10 000 users, every item a candidate) through the
    oracle's restatement of the reference pipeline -- one query after the other: score_explicit
    over all items, means added back, own items struck, argtopn (src/lenskit/knn/item.py:231-295,
    basic/candidates.py:77-94, basic/topn.py:45-69) -- on as many of the batch's queries as fit
    the budget, taken evenly across the (heaviest-first) batch: TIMED (cpu_baseline, extrapolated
    by similarity entries streamed) and COMPARED: sorted score rows bit for bit, every listed item
    really carrying its score and not one of the user's own.
    """
    import scipy.sparse as sps

    from oracle import lk_oracle as lko

    h = sps.csr_array((sims.values.cpu().numpy(), sims.indices.cpu().numpy(),
                       sims.indptr.cpu().numpy()), shape=sims.shape)
    threads = lko.num_threads()
    B = len(r_ptr) - 1
    pick = np.unique(np.linspace(0, B - 1, min(B, 2048)).astype(np.int64))
    done, secs, wi, ws, rows_ = [], 0.0, [], [], []
    for c0 in range(0, len(pick), 128):
        qs = pick[c0:c0 + 128]
        ptr = np.zeros(len(qs) + 1, np.int64)
        np.cumsum(r_ptr[qs + 1] - r_ptr[qs], out=ptr[1:])
        take = np.concatenate([np.arange(r_ptr[q], r_ptr[q + 1]) for q in qs])
        t0 = time.perf_counter()
        a, b, c = lko.iknn_recommend_batch(h, ptr, r_idx[take], r_val[take], means, 100, 1, n,
                                           threads, chunk=128)
        secs += time.perf_counter() - t0
        wi.append(a)
        ws.append(b)
        rows_.append(c)
        done.extend(qs.tolist())
        if secs > budget_s:
            break
    done = np.asarray(done)
    wi, ws, rows_ = np.concatenate(wi), np.concatenate(ws), np.concatenate(rows_)
    gi, gs = got_i[done], got_s[done]
    sc_same = bool(np.array_equal(gs.view(np.uint32), ws.view(np.uint32)))
    bad = ties = 0
    for r, q in enumerate(done):
        g = gi[r][gi[r] >= 0]
        genuine = np.array_equal(rows_[r][g].view(np.uint32), gs[r][: len(g)].view(np.uint32)) \
            and len(np.unique(g)) == len(g) and len(g) == int((wi[r] >= 0).sum())
        if not genuine:
            bad += 1
        elif not np.array_equal(gi[r], wi[r]):
            ties += 1
    frac = float(hits[done].sum()) / max(float(hits.sum()), 1.0)
    return {
        "cpu_baseline": {"value": round(secs / max(frac, 1e-12), 2), "unit": "s", "cores": threads,
                         "kind": "port",
                         "sample": f"{len(done)} of {B} queries ({frac * 100:.1f}% of the "
                         f"similarity entries) in {secs:.2f}s (score every item + means + own "
                         f"items struck + heap top-{n} per query; scoring spread over {threads} "
                         "threads), extrapolated by entries"},
        "parity": {"queries_checked": int(len(done)), "list_length": int(n),
                   "score_rows_bit_identical": sc_same,
                   "lists_identical": int(len(done) - bad - ties),
                   "lists_differing_among_equal_scores": int(ties),
                   "mismatched_users": int(bad), "ok": bool(sc_same and bad == 0)},
    }


def cfg5_run(args, dev, world, rank, steps, warmup, topk_users=0):
    """
    BASELINE.json configs[4] / SURVEY.md 8d "cfg5 concrete input": U = 10^7, I = 10^6,
    nnz = 10^8 generated in HBM from seed 5 (Philox; csrc/synth.hip), values 40, als-implicit
    k = 256 (exact solver), ``steps`` timed epochs after ``warmup``; dense top-100 for a user
    slice x ALL items with the history excluded.  ``args.scale`` shrinks users, items and nnz
    together.  With world > 1 the users / items are row-sharded as for cfg2; on one GPU the
    whole problem runs on that GPU.  Returns the result object (a bench line of its own under
    ``--config cfg5``, the ``cfg5`` leg of the default line otherwise).
    """
    import torch
    import torch.distributed as dist

    from lkpy_amd import _device as D
    from lkpy_amd import _native, synth
    from lkpy_amd._als_engine import HipBackend, ImplicitALSEngine

    k = args.k if args.k not in (64, 128) else 256
    reg = 0.1
    c = synth.CFG5
    n_users = max(1024, int(c["n_users"] * args.scale))
    n_items = max(1024, int(c["n_items"] * args.scale))
    nnz = int(c["nnz"] * args.scale)
    t0 = time.perf_counter()
    csr = synth.zipf_csr_on_device(dev, n_users, n_items, nnz, seed=c["seed"], value=40.0)
    torch.cuda.synchronize(dev)
    gen_seconds = time.perf_counter() - t0
    backend = HipBackend(k, dev, _native.SOLVER_AUTO)
    t0 = time.perf_counter()
    eng = ImplicitALSEngine(csr, k, reg, reg, None, None, backend)
    torch.cuda.synchronize(dev)
    setup_seconds = time.perf_counter() - t0

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(warmup):
        eng.train_epoch()
    eng.check()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):  # no instrumentation inside the timed region
        du, di = eng.train_epoch()
    barrier()
    elapsed = time.perf_counter() - t0
    eng.check()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    try:
        coll = collective_times(eng, min(steps, 3), dev, world)
    except Exception as exc:  # noqa: BLE001
        coll = {"error": f"{type(exc).__name__}: {exc}"}
    # kernel times: one more epoch with HIP events on the launch stream
    eng.u_plan.enable_timing(True)
    eng.i_plan.enable_timing(True)
    eng.train_epoch()
    barrier()
    cu, su, nu = eng.u_plan.get_timing()
    ci, si, ni = eng.i_plan.get_timing()
    eng.u_plan.enable_timing(False)
    eng.i_plan.enable_timing(False)
    ulen = np.diff(eng.u_plan.csr.h_indptr)
    ilen = np.diff(eng.i_plan.csr.h_indptr)
    uwb, iwb = bool(eng.u_plan.use_wb), bool(eng.i_plan.use_wb)
    fu, fuc = half_flops(ulen, k, uwb)
    fi, fic = half_flops(ilen, k, iwb)
    # per epoch: both solve launches + both chunk launches
    ep_ms = (su + si + cu + ci) / max(nu, 1)
    ep_flops = fu + fi + fuc + fic
    ex_flops = half_mfma_flops(ulen, backend.kp, uwb) + half_mfma_flops(ilen, backend.kp, iwb)
    out = {
        "metric": "ALS-implicit epochs/sec (cfg5 synthetic %.3gM x %.3gM x %.3gM, k=%d)"
        % (n_users / 1e6, n_items / 1e6, nnz / 1e6, k),
        "value": round(steps / elapsed, 4),
        "unit": "epochs/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (generated in HBM: lkpy_amd.synth.zipf_csr_on_device, Philox seed %d; "
        "degrees truncated-Zipf mean %.1f, items ~ Zipf(1.0))" % (c["seed"], nnz / n_users),
        "config": {
            "workload": "cfg5: %d users x %d items x %d interactions, als-implicit k=%d, "
            "%d timed epochs, %d x MI355X" % (n_users, n_items, nnz, k, steps, world),
            "solver": "cholesky" if eng.u_plan.solver == 0 else "cg",
            "reg": reg, "weight": 40.0,
            "longest_user_row": int(ulen.max()), "busiest_item": int(ilen.max()),
            "parallelism": ("row-sharded x%d, %d row slices per half-epoch (gathers under the solve)"
                            % (world, getattr(eng, "slices", 1))) if world > 1 else "single GPU",
        },
        "final_deltas": [float(du.item()), float(di.item())],
        "generate_seconds": round(gen_seconds, 3),
        "setup_seconds": round(setup_seconds, 3),
        "roofline": {
            "kernel": "als_blk_solve_kernel%d + als_blk_chunk_kernel" % (backend.kp // 16)
            + (" + als_wb / als_wb64 / als_wb128 kernels (rows <= 128 entries)" if (uwb or iwb) else ""),
            "bound": "mfma",
            "achieved": round(ep_flops / (ep_ms * 1e-3) / 1e12, 3),
            "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ep_flops / (ep_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
            "frac_executed": round(ex_flops / ((su + si) / max(nu, 1) * 1e-3) / 1e12
                                   / F32_MFMA_PEAK_TFLOPS, 4),
            "kernel_ms_per_epoch": {"user_solve": round(su / max(nu, 1), 3),
                                    "item_solve": round(si / max(ni, 1), 3),
                                    "user_chunk": round(cu / max(nu, 1), 3),
                                    "item_chunk": round(ci / max(ni, 1), 3)},
            "algorithmic_flops_per_epoch": ep_flops,
            "reference_flops_per_epoch": reference_half_flops(ulen, k)
            + reference_half_flops(ilen, k),
            "woodbury_rows": {"user": int(eng.u_plan.woodbury_rows) if uwb else 0,
                              "item": int(eng.i_plan.woodbury_rows) if iwb else 0},
            "algorithmic_bytes_per_epoch": half_bytes(ulen, k) + half_bytes(ilen, k),
            "traffic": None, "traffic_source": None,
            "note": "rows with <= 128 entries are rank-n updates of OtOr and are solved through "
            "the Woodbury identity (same solution, O(n^2 k) flops): algorithmic_flops counts "
            "what this path needs for them; reference_flops = a dense k^3/3 solve for every "
            "row, as the reference does (SURVEY 8d)",
        },
    }

    if world == 1 and args.scale == 1.0:
        # HBM bytes per epoch over ALL kernels of the epoch, from the committed PMC summary
        out["roofline"]["traffic"], out["roofline"]["traffic_source"] = pmc_traffic_per_epoch(
            "r*_cfg5_counters.csv", "als_blk_solve_kernel16")
    if coll:
        out["collectives"] = coll
    if world == 1 and not NO_ORDER_AB:
        out["summation_order"] = summation_order_info(
            eng, lambda mode: ImplicitALSEngine(csr, k, reg, reg, None, None,
                                                HipBackend(k, dev, _native.SOLVER_AUTO, mode)),
            steps, elapsed / steps * 1e3, dev)

    def leg(name, fn):
        try:
            out[name] = fn()
        except Exception as exc:  # noqa: BLE001 -- reported, not swallowed
            out[name] = {"error": f"{type(exc).__name__}: {exc}"}

    def parity_leg():
        # "cfg5 on CPU: time 1 half-epoch on a 1 % row sample and extrapolate"; the
        # same samples are the parity check -- BOTH halves from identical inputs: sampled user
        # rows, and sampled item rows that always include the BUSIEST item (1.5 M entries at full
        # scale: the chunk path's hardest case; relabelled row 0)
        import scipy.sparse as sps

        from oracle import lk_oracle as lko
        from oracle import parity

        threads = lko.num_threads()
        rng = np.random.default_rng(3)
        res, cpu_s, cpu_fl, desc = {}, 0.0, 0.0, []

        def half(name, plan, this_full, lo, hi, other_full, otor_reg, n_sample, must=()):
            nonlocal cpu_s, cpu_fl
            other_h = eng.backend.download(other_full)  # relabelled order on both sides
            n_rows = plan.csr.shape[0]
            rows = rng.choice(n_rows, min(n_rows, n_sample), replace=False)
            rows = np.unique(np.concatenate([rows, np.asarray(must, dtype=rows.dtype)]))
            hp = plan.csr.h_indptr.astype(np.int64)
            lens = hp[rows + 1] - hp[rows]
            ptr = np.zeros(len(rows) + 1, np.int64)
            np.cumsum(lens, out=ptr[1:])
            # (plans are views into the full arrays: offsets are not rebased)
            take = torch.from_numpy(np.concatenate(
                [np.arange(hp[r], hp[r + 1]) for r in rows]).astype(np.int64)).to(dev)
            idx = plan.csr.indices[take].cpu().numpy()
            val = plan.csr.values[take].cpu().numpy()
            sub = sps.csr_array((val, idx, ptr), shape=(len(rows), other_h.shape[0]))
            # the GPU's half from these inputs: one more half-epoch, sampled rows read back
            otor = eng.backend.gramian(other_full, otor_reg)
            eng.backend.half_epoch(plan, this_full[lo:hi], other_full, otor)
            plan.check_status()
            got = eng.backend.download(this_full[lo:hi][torch.from_numpy(rows).to(dev)])
            want = np.zeros_like(got)
            t0 = time.perf_counter()
            otor_h = lko.implicit_otor(other_h, otor_reg)
            lko.als_half_epoch(sub, want, other_h, otor_h, threads)
            dt = time.perf_counter() - t0
            exact, cond = lko.als_referee_f64(sub, other_h, otor_reg)
            acc = parity.als_half_accounting(got, want, exact, cond)
            acc.pop("by_cond_decade", None)
            acc["longest_row_checked"] = int(lens.max())
            if acc.get("exceptions"):
                # reproduce them: the same half with the rhs in the reference's summation order
                ex = np.array([e["row"] for e in acc["exceptions"]])
                for e in acc["exceptions"]:
                    e["entries"] = int(lens[e["row"]])
                acc["exceptions_in_reference_order"] = reference_order_recheck(
                    sps.csr_array(sub[ex]), other_full, otor_h, want[ex], k, dev)
            cpu_s += dt
            cpu_fl += reference_half_flops(lens, k)
            desc.append(f"{name} half: {len(rows)} of {n_rows} rows ({sub.nnz} nnz, longest "
                        f"{int(lens.max())}) in {dt:.2f}s")
            res[name] = acc

        n_u = max(256, eng.u_plan.csr.shape[0] // 400)
        n_i = max(256, eng.i_plan.csr.shape[0] // 400)
        # EVERY row of more than 4096 entries is checked (the rows where the
        # reference's float32 sums drift: ~2 600 items at full scale, 60 % of the entries), not
        # just the busiest one; plus the seeded 0.25 % sample of the others
        def long_rows(plan):
            hp = plan.csr.h_indptr.astype(np.int64)
            return np.flatnonzero(np.diff(hp) > 4096)

        lu, li = long_rows(eng.u_plan), long_rows(eng.i_plan)
        half("user", eng.u_plan, eng.P, eng.u_lo, eng.u_hi, eng.Q, reg, n_u, must=lu)
        half("item", eng.i_plan, eng.Q, eng.i_lo, eng.i_hi, eng.P, reg, n_i, must=li)
        res["user"]["rows_over_4096_entries_checked"] = int(len(lu))
        res["item"]["rows_over_4096_entries_checked"] = int(len(li))
        # keep the engine consistent for the legs that follow (the scorer's Q^T Q)
        eng._qtq = eng.backend.gramian(eng.Q, reg)
        est = cpu_s * (reference_half_flops(ulen, k) + reference_half_flops(ilen, k)) \
            / max(cpu_fl, 1.0)
        out["cpu_baseline"] = {
            "value": 1.0 / est, "unit": "epochs/s", "cores": threads, "kind": "port",
            "sample": "; ".join(desc) + "; extrapolated to an epoch by algorithmic flops "
            "(reference's dense k^3/3 per row)",
            "host_cpus": os.cpu_count()}
        reco = [o["exceptions_in_reference_order"] for o in res.values()
                if "exceptions_in_reference_order" in o]
        return {"what": "user AND item half-epoch, GPU vs oracle from identical inputs: EVERY row of "
                "more than 4096 entries + a seeded 0.25 % sample of the others",
                "rows_over_4096_entries_checked": int(len(lu) + len(li)),
                "rows_checked": res["user"]["rows"] + res["item"]["rows"],
                "rows_over_1e-4": res["user"]["rows_over_1e-4"] + res["item"]["rows_over_1e-4"],
                "row_rel_max": max(res["user"]["row_rel_max"], res["item"]["row_rel_max"]),
                "ok": bool(res["user"]["ok"] and res["item"]["ok"]),
                "ok_in_reference_order": bool(all(r["within_1e-4"] == r["rows"] for r in reco)),
                "exceptions_reproduced_in_reference_order": [
                    sum(r["within_1e-4"] for r in reco), sum(r["rows"] for r in reco)],
                "accounted": bool(res["user"]["accounted"] and res["item"]["accounted"]),
                "exceptions": [dict(e, half=h) for h in res for e in res[h].get("exceptions", [])],
                "user": res["user"], "item": res["item"]}

    def topk_leg():
        B = topk_users or max(64, eng.P.shape[0] // 8)
        B = min(B, eng.P.shape[0])
        hp = eng.u_plan.csr.h_indptr.astype(np.int64)
        excl_ptr = torch.from_numpy(hp[: B + 1]).to(dev)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        g_idx, g_sc = D.score_topk(eng.P[:B], eng.Q, k, 100, excl_ptr, eng.u_plan.csr.indices)
        torch.cuda.synchronize(dev)
        tb = time.perf_counter() - t0
        fl = 2.0 * B * eng.Q.shape[0] * k
        res = {"metric": "dense scoring + top-100, %d users x %d items (k=%d), seconds"
               % (B, eng.Q.shape[0], k),
               "value": round(tb, 3), "unit": "s", "users_per_s": round(B / tb, 1),
               "roofline": {"bound": "mfma", "achieved": round(fl / tb / 1e12, 2),
                            "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(fl / tb / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                            "algorithmic_flops": fl}}
        if not args.no_cpu:
            # the reference's per-query path on a seeded user sample of THIS slice: timed and
            # compared (index lists + score bits) -- VERDICT r3: the slice had neither
            try:
                n_ex = int(hp[B])
                res["cpu_baseline"], res["parity"] = topk_cpu_and_parity(
                    eng.backend.download(eng.P[:B]), eng.backend.download(eng.Q), hp[: B + 1],
                    eng.u_plan.csr.indices[:n_ex].cpu().numpy(), g_idx.cpu().numpy(),
                    g_sc.cpu().numpy(), 100, budget_s=8.0, max_users=4096, block=128)
            except Exception as exc:  # noqa: BLE001 -- reported in place
                res["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"}
        return res

    if rank == 0 and world == 1 and not args.no_cpu:
        leg("parity", parity_leg)
    if rank == 0 and world == 1 and not args.no_topk:
        leg("topk", topk_leg)
    del eng
    torch.cuda.empty_cache()
    return out



def init_ranks(world: int, local_rank: int):
    """
    Device + process group of this rank: ``nccl`` (= RCCL) with one GPU per rank -- the contract.
    ``LK_BENCH_DRYRUN_ONE_GPU=1`` (a rehearsal, never a measurement): every rank uses cuda:0 and
    the group is ``gloo`` moving device tensors, so that the whole N > 1 flow of this file
    (sharded engine, collective timing, predicted-vs-measured, sharded legs, watchdog) can be
    executed on a one-GPU box before an 8-GPU node sees it; the line says ``"dryrun": true``.
    """
    import torch
    import torch.distributed as dist

    dry = os.environ.get("LK_BENCH_DRYRUN_ONE_GPU", "0") == "1"
    idx = 0 if dry else local_rank
    torch.cuda.set_device(idx)
    dev = torch.device("cuda", idx)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    return dev, dry


def main_cfg5(args):
    "``--config cfg5``: the cfg5 run as a bench line of its own."
    import torch
    import torch.distributed as dist

    from lkpy_amd import _native

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    _native.require_gpu()
    dev, dry = init_ranks(world, local_rank)
    out = cfg5_run(args, dev, world, rank, args.steps, args.warmup, args.topk_users)
    if dry and world > 1:
        out["dryrun"] = True
    if rank == 0:
        emit(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def summation_order_info(eng, make_engine, steps, ms_default, dev):
    """
    VERDICT r4 item 1: the epoch time of the new DEFAULT summation order (hybrid: rows of more than
    LK_ALS_REF_LEN entries in the reference's own order) beside round 4's default
    (``LK_ALS_RHS_ORDER=accurate``) on the same inputs.  ``make_engine(mode)`` builds an engine of
    the same problem in that mode; it is timed over ``steps`` epochs after 2 warm-up epochs.
    """
    import torch

    info = {
        "default": getattr(eng.u_plan, "order_mode", None),
        "rows_in_reference_order": {"user": int(eng.u_plan.long_rows()),
                                    "item": int(eng.i_plan.long_rows())},
        "ref_len": int(os.environ.get("LK_ALS_REF_LEN", "2048")),
        "ms_per_step": round(ms_default, 4),
    }
    try:
        e2 = make_engine("accurate")
        for _ in range(2):
            e2.train_epoch()
        e2.check()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            e2.train_epoch()
        torch.cuda.synchronize(dev)
        info["ms_per_step_accurate_order"] = round((time.perf_counter() - t0) / steps * 1e3, 4)
        info["cost_of_reference_order"] = round(ms_default / info["ms_per_step_accurate_order"], 4)
        del e2
        torch.cuda.empty_cache()
    except Exception as exc:  # noqa: BLE001 -- reported in place
        info["error"] = f"{type(exc).__name__}: {exc}"
    return info


# one-GPU epoch times (ms) of the committed bench lines, and the per-epoch time of work that does
# NOT shrink with the ranks (cfg5's Z = other @ OtOr^-1 GEMM + inverse, every rank in full unless
# LK_ALS_Z=sharded) -- the inputs of DESIGN.md section 6's predicted table
ONE_GPU_MS = {64: 3.59, 128: 15.3, 256: 202.0}
REPLICATED_MS = {256: 14.0}


def predicted_epoch(eng, world):
    """
    DESIGN.md section 6's model of an N-GPU epoch, evaluated for THIS run so the measured line can
    be read against it without a calculator: sharded kernels = one-GPU epoch / N; every rank's
    replicated work; the row gathers on a full mesh ((N - 1)/N of the bytes over min(N - 1, 7)
    links of 153 GB/s), of which only the last of S row slices is exposed; 2 small all-reduces of
    ~30 us.  ``ring_ms``: the same bytes if RCCL rings (one link's rate).  A prediction made before
    any multi-GPU run existed -- its value is in how the measured terms differ from it.
    """
    try:
        kp = int(getattr(eng.backend, "kp", eng.k))
        n1 = ONE_GPU_MS.get(kp)
        if n1 is None or world < 2:
            return None
        from lkpy_amd._als_engine import sharded_z
        S = int(getattr(eng, "slices", 1))
        gathered = (eng.P.shape[0] + eng.Q.shape[0]) * kp * 4.0
        zrep = REPLICATED_MS.get(kp, 0.0)
        if sharded_z():
            gathered += (eng.P.shape[0] + eng.Q.shape[0]) * kp * 4.0 if zrep else 0.0
            zrep = zrep / world
        link = 153e9
        mesh = (world - 1) / world * gathered / (min(world - 1, 7) * link) * 1e3
        ring = (world - 1) / world * gathered / link * 1e3
        sharded = (n1 - REPLICATED_MS.get(kp, 0.0)) / world
        fixed = 0.06
        return {"model": "DESIGN.md section 6", "one_gpu_ms": n1, "world": world,
                "sharded_kernels_ms": round(sharded, 3), "replicated_ms": round(zrep, 3),
                "gather_bytes_per_epoch": gathered, "row_slices": S,
                "gather_mesh_ms": round(mesh, 3), "gather_exposed_ms": round(mesh / S, 3),
                "small_all_reduces_ms": fixed,
                "epoch_ms": round(sharded + zrep + mesh / S + fixed, 3),
                "epoch_ms_if_ring": round(sharded + zrep + max(ring / S, ring - sharded) + fixed, 3),
                "z": "sharded" if sharded_z() else "replicated"}
    except Exception as exc:  # noqa: BLE001 -- a reading aid, never costs the line
        return {"error": f"{type(exc).__name__}: {exc}"}


def collective_times(eng, steps, dev, world):
    """
    world > 1: ``steps`` more epochs with the engine's collectives bracketed by events
    (``TorchComm.enable_timing``): milliseconds per epoch a rank's stream spent in (or waiting
    for) each kind of collective, for EVERY rank -- gathered to rank 0.  Not part of the timed
    region; the numbers to hold against DESIGN.md section 6's predicted table.
    """
    import torch
    import torch.distributed as dist

    comm = getattr(eng, "comm", None)
    if world <= 1 or comm is None or not hasattr(comm, "enable_timing"):
        return None
    comm.enable_timing(True)
    torch.cuda.synchronize(dev)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.train_epoch()
    torch.cuda.synchronize(dev)
    wall = (time.perf_counter() - t0) / steps * 1e3
    mine = {k_: round(v / steps, 4) for k_, v in comm.timing_ms().items()}
    mine["epoch_ms_with_events"] = round(wall, 4)
    comm.enable_timing(False)
    allr = [None] * world
    dist.all_gather_object(allr, mine)
    return {"per_rank_ms_per_epoch": allr,
            "predicted": predicted_epoch(eng, world),
            "note": "stream time inside blocking collectives / exposed wait of the asynchronous "
                    "row gathers, per epoch, rank by rank (events on the launch stream; a "
                    "separate pass after the timed region)"}


def als_timed(ui, P0, Q0, k, reg, steps, warmup, dev, world, scale):
    """
    Set up the engine (one upload, relabel + transpose in HBM), run ``warmup`` untimed and
    ``steps`` timed epochs bracketed by barrier + synchronize (maximum over ranks), and build the
    roofline object of the solve kernel from the HIP-event launch durations recorded on the
    launch stream inside the library.  Returns (engine, backend, seconds, roofline, set-up
    seconds, final deltas).
    """
    import torch
    import torch.distributed as dist

    from lkpy_amd import _native
    from lkpy_amd._als_engine import HipBackend, ImplicitALSEngine

    backend = HipBackend(k, dev, _native.SOLVER_AUTO)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    eng = ImplicitALSEngine(ui, k, reg, reg, P0, Q0, backend)
    torch.cuda.synchronize(dev)
    setup_seconds = time.perf_counter() - t0

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(warmup):
        eng.train_epoch()
    eng.check()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):  # the headline: no instrumentation inside the timed region
        du, di = eng.train_epoch()
    barrier()
    elapsed = time.perf_counter() - t0
    eng.check()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    try:
        eng.collectives = collective_times(eng, min(steps, 10), dev, world)
    except Exception as exc:  # noqa: BLE001 -- reported in place, never costs the headline
        eng.collectives = {"error": f"{type(exc).__name__}: {exc}"}

    # ---- roofline of the dominant kernel (local shard of this rank): a SEPARATE pass of the
    # same epochs with HIP events around the kernels on the launch stream ----
    roof = None
    eng.u_plan.enable_timing(True)
    eng.i_plan.enable_timing(True)
    for _ in range(min(steps, 20)):
        eng.train_epoch()
    barrier()
    cu, su, nu = eng.u_plan.get_timing()
    ci, si, ni = eng.i_plan.get_timing()
    eng.u_plan.enable_timing(False)
    eng.i_plan.enable_timing(False)
    ulen = np.diff(eng.u_plan.csr.h_indptr)
    ilen = np.diff(eng.i_plan.csr.h_indptr)
    uwb = bool(getattr(eng.u_plan, "use_wb", False))
    iwb = bool(getattr(eng.i_plan, "use_wb", False))
    fu_solve, fu_chunk = half_flops(ulen, k, uwb)
    fi_solve, fi_chunk = half_flops(ilen, k, iwb)
    launches = nu + ni
    kname = ("als_solve_kernel<NT=%d>" if backend.kp <= 64 else "als_blk_solve_kernel%d") % (
        backend.kp // 16)
    # LK_ALS_FUSED=1 (an experiment of round 4, slower, off by default): the chunk blocks and the
    # solve blocks of the rows without chunks run in ONE launch (als_fused_kernel), followed by a
    # small als_solve_kernel launch for the rows that consume the slabs: the plan's two timers are
    # then (fused launch, that small launch) and the roofline is taken over both
    kmatch = kname.split("<")[0]  # (the name the PMC summaries are searched for)
    if uwb or iwb:  # (the plan's solve timer brackets these launches too)
        kname += " + als_wb / als_wb64 kernels of the short rows (same HIP-event bracket)"
    fused = backend.kp <= 64 and os.environ.get("LK_ALS_FUSED", "0") == "1" and (cu + ci) > 0
    if fused:
        kname = "als_fused_kernel<NT=%d> (chunk + solve blocks interleaved) + als_solve_kernel " \
                "(rows with chunks)" % (backend.kp // 16)
    if launches > 0 and (su + si) > 0:
        flops_per_launch = (fu_solve * nu + fi_solve * ni) / launches
        exec_per_launch = (half_mfma_flops(ulen, backend.kp, uwb) * nu
                           + half_mfma_flops(ilen, backend.kp, iwb) * ni) / launches
        avg_ms = (su + si) / launches
        if fused:
            flops_per_launch += (fu_chunk * nu + fi_chunk * ni) / launches
            # (chunk blocks issue the upper tiles too: (NT + 1) / (2 NT) of the 2 k^2 convention)
            nt = backend.kp // 16
            exec_per_launch += (fu_chunk * nu + fi_chunk * ni) / launches * (nt + 1) / (2.0 * nt)
            avg_ms = (su + si + cu + ci) / launches
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
        roof = {
            "kernel": kname,
            "bound": "mfma",
            "achieved": round(achieved, 3),
            "peak": F32_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 4),
            "frac_executed": round(exec_per_launch / (avg_ms * 1e-3) / 1e12
                                   / F32_MFMA_PEAK_TFLOPS, 4),
            "traffic": None,
            "traffic_source": None,
            "avg_launch_ms": round(avg_ms, 4),
            "launches": launches,
            "algorithmic_flops_per_launch": flops_per_launch,
            "executed_flops_per_launch": exec_per_launch,
            "algorithmic_bytes_per_launch": (half_bytes(ulen, k) * nu + half_bytes(ilen, k) * ni)
            / launches,
            "chunk_kernel_ms_per_launch": None if fused else round((cu + ci) / launches, 4),
            "fused_launch_ms": round((cu + ci) / launches, 4) if fused else None,
            "long_row_solve_ms": round((su + si) / launches, 4) if fused else None,
            "chunk_kernel_flops_per_launch": (fu_chunk * nu + fi_chunk * ni) / launches,
            "note": "frac = SURVEY 8d algorithmic flops (2k^2 per entry) / time / peak; "
            "frac_executed = matrix-core work actually issued (upper tiles only)"
            + ("; rows with <= 64 entries take the Woodbury kernels and are counted with that "
               "method's flops (user %d, item %d rows)"
               % (eng.u_plan.woodbury_rows if uwb else 0, eng.i_plan.woodbury_rows if iwb else 0)
               if (uwb or iwb) else ""),
        }
    if roof and world == 1 and scale == 1.0:
        roof["traffic"], roof["traffic_source"] = pmc_traffic(
            "r*_k%d_counters.csv" % k if k != 64 else "r*_als_*_counters.csv",
            kmatch)
        if roof["traffic"] is None and fused:  # (captures made before the fused launch)
            roof["traffic"], roof["traffic_source"] = pmc_traffic(
                "r*_als_*_counters.csv", "als_solve_kernel")
    if world == 1 and not NO_ORDER_AB:
        eng.summation_order = summation_order_info(
            eng, lambda mode: ImplicitALSEngine(ui, k, reg, reg, P0, Q0,
                                                HipBackend(k, dev, _native.SOLVER_AUTO, mode)),
            min(steps, 20), elapsed / steps * 1e3, dev)
    return eng, backend, elapsed, roof, setup_seconds, (float(du.item()), float(di.item()))


def _pick(d, *keys):
    "the named keys of a dict that are present (compact line)"
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _short(text, n=160):
    text = str(text)
    return text if len(text) <= n else text[: n - 3] + "..."


def _roof(r):
    return _pick(r, "kernel", "bound", "achieved", "peak", "unit", "frac", "frac_executed",
                 "traffic", "avg_launch_ms") if isinstance(r, dict) else None


def _cpu(c):
    if not isinstance(c, dict):
        return None
    d = _pick(c, "value", "unit", "cores", "kind", "error")
    if isinstance(c.get("threads_tried_user_half_seconds"), dict) and c["threads_tried_user_half_seconds"]:
        d["threads_tried_s"] = c["threads_tried_user_half_seconds"]
    if "sample" in c:
        d["sample"] = _short(c["sample"], 140)
    return d


def _als_par(par):
    "digest of an ALS parity object: the raw criterion, counts, the exception rows"
    if not isinstance(par, dict):
        return None
    d = _pick(par, "ok", "rows_checked", "rows_over_1e-4", "row_rel_max", "accounted",
              "rows_over_4096_entries_checked", "ok_in_reference_order",
              "exceptions_reproduced_in_reference_order", "error")
    if par.get("exceptions"):
        d["exceptions"] = [{k: (round(v, 8) if isinstance(v, float) else v) for k, v in
                            _pick(e, "half", "row", "entries", "cond", "gpu_vs_oracle",
                                  "gpu_vs_f64", "oracle_vs_f64").items()}
                           for e in par["exceptions"][:6]]
    return d


def compact_line(out: dict) -> dict:
    """
    The LAST line bench.py prints: the contract's headline fields + ``roofline`` +
    ``cpu_baseline`` + a digest of every leg (value, roofline fraction, CPU baseline, parity
    verdict with counts and exceptions), small enough (< 6 KB) for a driver that keeps only the
    tail of stdout.  The full objects are in the line printed just before it.
    """
    c = _pick(out, "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
              "higher_is_better", "scaling", "vs_baseline", "dtype", "collectives", "incomplete",
              "dryrun")
    c["data"] = _short(out.get("data", ""), 200)
    c["config"] = _pick(out.get("config", {}), "workload", "solver", "parallelism", "data_source")
    if "roofline" in out:
        c["roofline"] = _roof(out["roofline"])
    if "cpu_baseline" in out:
        c["cpu_baseline"] = _cpu(out["cpu_baseline"])
    if isinstance(out.get("summation_order"), dict):
        c["summation_order"] = _pick(out["summation_order"], "default", "ms_per_step",
                                     "ms_per_step_accurate_order", "rows_in_reference_order")
    par = out.get("parity")
    if isinstance(par, dict):
        c["parity"] = _als_par(par)
        if isinstance(par.get("knn"), dict):
            c["parity"]["knn_build"] = par["knn"]
    legs = {}
    knn = out.get("knn")
    if isinstance(knn, dict):
        d = _pick(knn, "value", "unit", "build_seconds_hbm_resident",
                  "build_seconds_to_host_first_call", "build_save_nbrs_100_seconds",
                  "prepare_seconds", "nnz_out", "error")
        d["metric"] = ("item-kNN model build seconds, CSR on device -> similarity CSR on host "
                       "(BASELINE's definition; hbm_resident = without the download)"
                       if "build_seconds_hbm_resident" in knn else knn.get("metric"))
        d["roofline"] = _roof(knn.get("roofline"))
        d["cpu_baseline"] = _cpu(knn.get("cpu_baseline"))
        for name in ("batch_score", "recommend"):
            sub = knn.get(name)
            if isinstance(sub, dict):
                e = _pick(sub, "seconds", "queries", "queries_per_s", "targets_per_query", "n",
                          "error")
                e["roofline_frac"] = (sub.get("roofline") or {}).get("frac")
                if isinstance(sub.get("through_components"), dict):
                    e["seconds_through_components"] = sub["through_components"].get("seconds")
                e["cpu_baseline"] = _cpu(sub.get("cpu_baseline"))
                e["parity"] = _pick(sub.get("parity") or {}, "ok", "queries_checked",
                                    "scores_compared", "scores_bit_identical", "counts_identical",
                                    "lists_identical", "mismatched_users", "error")
                d[name] = e
        legs["knn"] = d
    topk = out.get("topk")
    if isinstance(topk, dict):
        d = _pick(topk, "metric", "value", "unit", "users_per_s", "error")
        d["roofline"] = _roof(topk.get("roofline"))
        d["cpu_baseline"] = _cpu(topk.get("cpu_baseline"))
        d["parity"] = topk.get("parity")
        legs["topk"] = d
    if isinstance(out.get("fit"), dict):
        legs["fit"] = _pick(out["fit"], "fit_seconds", "epochs_per_s_from_log",
                            "setup_and_download_seconds", "error")
    rec = out.get("recommend")
    if isinstance(rec, dict):
        d = _pick(rec, "value", "unit", "users", "users_per_s", "seconds_through_batch_recommend",
                  "device_ms", "host_fraction_of_call", "error")
        d["metric"] = "als-implicit.toml batch recommend (history -> fold-in -> score -> top-100)"
        d["roofline"] = _roof(rec.get("roofline"))
        d["cpu_baseline"] = _cpu(rec.get("cpu_baseline"))
        pr = rec.get("parity") or {}
        d["parity"] = {"ok": pr.get("ok"),
                       "fold_in": _pick(pr.get("fold_in") or {}, "rows_checked", "rows_over_1e-4",
                                        "row_rel_max", "ok"),
                       "lists": _pick(pr.get("lists") or {}, "users_checked",
                                      "score_rows_bit_identical", "lists_identical",
                                      "mismatched_users", "ok")}
        legs["recommend"] = d
    if isinstance(out.get("cg"), dict):
        legs["cg"] = _pick(out["cg"], "exact_ms_per_epoch", "cg_ms_per_epoch",
                           "cg_iterations_per_row", "one_epoch_rel_diff_P", "one_epoch_rel_diff_Q",
                           "user_rows", "item_rows", "error")
    for name in ("k128", "cfg5"):
        leg = out.get(name)
        if not isinstance(leg, dict):
            continue
        d = _pick(leg, "metric", "value", "unit", "ms_per_step", "steps", "setup_seconds", "error")
        d["roofline"] = _roof(leg.get("roofline"))
        d["cpu_baseline"] = _cpu(leg.get("cpu_baseline"))
        d["parity"] = _als_par(leg.get("parity"))
        if isinstance(leg.get("summation_order"), dict):
            d["summation_order"] = _pick(leg["summation_order"], "default", "ms_per_step",
                                         "ms_per_step_accurate_order")
        if isinstance(leg.get("topk"), dict):
            t = leg["topk"]
            d["topk"] = _pick(t, "value", "unit", "users_per_s", "error")
            d["topk"]["roofline_frac"] = (t.get("roofline") or {}).get("frac")
            d["topk"]["cpu_baseline"] = _cpu(t.get("cpu_baseline"))
            d["topk"]["parity"] = _pick(t.get("parity") or {}, "ok", "users_checked",
                                        "score_rows_bit_identical", "lists_identical",
                                        "mismatched_users")
        legs[name] = d
    for name in ("sharded_topk", "sharded_knn", "sharded_legs_error"):
        if name in out:
            legs[name] = out[name]
    if legs:
        c["legs"] = legs
    c["full_line"] = "the JSON line printed before this one carries every leg in full"
    return c


def _sig(x, digits=5):
    "floats to 5 significant digits, recursively (the compact line only)"
    if isinstance(x, float):
        return float(f"{x:.{digits}g}") if np.isfinite(x) else x
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def emit(out: dict):
    "rank 0: the full line first, then the compact line LAST (what a tail-keeping driver parses)"
    print(json.dumps(out), flush=True)
    c = _sig(compact_line(out))
    sep = (",", ":")
    line = json.dumps(c, separators=sep)
    # never let the digest outgrow the driver's tail (8000 characters): drop detail in this
    # order, keep every value / fraction / verdict
    def drop_samples(leg):
        for key, val in list(leg.items()):
            if isinstance(val, dict):
                if key == "cpu_baseline":
                    val.pop("sample", None)
                drop_samples(val)

    steps = [lambda: drop_samples(c.get("legs", {})),
             lambda: c.get("legs", {}).pop("cg", None),
             lambda: c.get("legs", {}).pop("fit", None),
             lambda: c.pop("data", None)]
    for step in steps:
        if len(line) <= 6000:
            break
        step()
        line = json.dumps(c, separators=sep)
    print(line, flush=True)


def load_ratings(scale: float):
    """
    SURVEY 8d / BASELINE.md "Inputs": the real MovieLens-25M when it is on the box -- the
    reference's fixture path ``data/ml-25m.zip`` (src/lenskit/testing/_movielens.py:34) or an
    unpacked ``data/ml-25m/``, relative to the repository or the working directory, or whatever
    ``LK_ML25M`` names -- read by ``lkpy_amd.data.load_movielens`` (the reference's
    ``load_movielens``: every movies.csv id is an item); else the seeded ML-25M-shaped synthetic.
    Returns (ratings CSR users x items, f32), a description for ``data``, the source tag.
    """
    from lkpy_amd import synth

    cands = [os.environ.get("LK_ML25M")] if os.environ.get("LK_ML25M") else []
    for base in (ROOT, Path.cwd()):
        cands += [base / "data" / "ml-25m.zip", base / "data" / "ml-25m"]
    for c in cands:
        c = Path(c)
        if scale == 1.0 and c.exists():
            from lkpy_amd.data import load_movielens

            ds = load_movielens(c)
            ratings = ds.interactions().matrix().scipy(attribute="rating", layout="csr")
            info = synth.describe(ratings)
            return ratings, ("MovieLens-25M (%s via lkpy_amd.data.load_movielens: %d users x %d "
                             "items, %d ratings, %d unrated items)"
                             % (c, info["n_users"], info["n_items"], info["nnz"],
                                info["empty_items"])), str(c)
    ratings = synth.ml25m_like(scale=scale)
    info = synth.describe(ratings)
    return ratings, ("synthetic (seeded ML-25M-shaped: lkpy_amd.synth.ml25m_like, seed 20260925; "
                     "the public dataset's counts exactly: nnz %d, user rows %d..%d, busiest item "
                     "%d, %d unrated items; no data/ml-25m.zip on this box)"
                     % (info["nnz"], info["user_len_min"], info["user_len_max"],
                        info["item_len_max"], info["empty_items"])), "synthetic"


def maybe_self_launch(args):
    """
    ``python bench.py --gpus N`` with N > 1 and no launcher environment: re-execute under
    ``torch.distributed.run`` (one process per GPU, rendezvous on 127.0.0.1 and a free port) --
    the same command line the driver uses when it launches the ranks itself, in which case
    WORLD_SIZE is already set and this is a no-op.
    """
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed epochs (default 50; cfg5: 3, SURVEY 8d).  BASELINE's 20-epoch "
                    "fit is the `fit` leg; more timed epochs only steady the epochs/s figure")
    ap.add_argument("--warmup", type=int, default=None, help="default 3 (cfg5: 1)")
    ap.add_argument("--k", type=int, default=64)
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the dataset (debug only)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the parity + cpu_baseline legs")
    ap.add_argument("--no-knn", action="store_true", help="skip the item-kNN build leg")
    ap.add_argument("--no-topk", action="store_true", help="skip the dense top-N scoring leg")
    ap.add_argument("--no-fit", action="store_true", help="skip the end-to-end fit leg")
    ap.add_argument("--no-recommend", action="store_true",
                    help="skip the als-implicit.toml batch recommend leg (needs the fit leg)")
    ap.add_argument("--no-cg", action="store_true", help="skip the CG-solver comparison leg")
    ap.add_argument("--no-k128", action="store_true", help="skip the k = 128 leg (configs[3])")
    ap.add_argument("--no-cfg5", action="store_true", help="skip the cfg5 leg (configs[4])")
    ap.add_argument("--no-order-ab", action="store_true",
                    help="skip timing LK_ALS_RHS_ORDER=accurate beside the default summation "
                         "order (profiling runs: keeps the kernel statistics to one mode)")
    ap.add_argument("--k128-steps", type=int, default=10, help="timed epochs of the k128 leg")
    ap.add_argument("--sharded-legs", action=argparse.BooleanOptionalAction, default=True,
                    help="N > 1: also time the sharded item-kNN build and dense top-N (on by "
                    "default since round 4; a failure there is reported in place and never "
                    "costs the headline; --no-sharded-legs skips them)")
    ap.add_argument("--sharded-legs-timeout", type=int, default=240,
                    help="N > 1: seconds after which a watchdog prints the headline line without "
                    "the sharded legs and ends the run (a rank failing alone would otherwise "
                    "leave the others in a collective)")
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg5"],
                    help="cfg2: ML-25M-shaped (the default, BASELINE.json configs[1..3]); "
                    "cfg5: synthetic 10M x 1M x 100M generated in HBM, k = 256 (configs[4])")
    ap.add_argument("--topk-users", type=int, default=0,
                    help="cfg5: users of the dense top-K leg (default: 1/8 of the users)")
    args = ap.parse_args()
    global NO_ORDER_AB
    NO_ORDER_AB = bool(args.no_order_ab)
    maybe_self_launch(args)
    if args.steps is None:
        args.steps = 3 if args.config == "cfg5" else 50
    if args.warmup is None:
        args.warmup = 1 if args.config == "cfg5" else 3
    if args.config == "cfg5":
        return main_cfg5(args)

    import torch
    import torch.distributed as dist

    from lkpy_amd import _native, synth
    from lkpy_amd._als_engine import HipBackend, ImplicitALSEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    _native.require_gpu()  # no CPU fallback: fail loudly without the HIP path
    dev, dry = init_ranks(world, local_rank)

    k, reg, weight = args.k, 0.1, 40.0
    ratings, data_desc, data_source = load_ratings(args.scale)
    info = synth.describe(ratings)
    import scipy.sparse as sps

    ui = sps.csr_array(
        (np.full(ratings.nnz, weight, dtype=np.float32), ratings.indices, ratings.indptr),
        shape=ratings.shape,
    )
    # the reference's init: item matrix first, then users, (N(0,1)*0.01)^2
    rng = np.random.default_rng(42)
    Q0 = rng.standard_normal((ui.shape[1], k), dtype=np.float32) * 0.01
    Q0 *= Q0
    P0 = rng.standard_normal((ui.shape[0], k), dtype=np.float32) * 0.01
    P0 *= P0

    eng, backend, elapsed, roof, setup_seconds, deltas = als_timed(
        ui, P0, Q0, k, reg, args.steps, args.warmup, dev, world, args.scale)
    ms_per_step = elapsed / args.steps * 1e3

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    out = {
        "metric": "ALS-implicit epochs/sec (ML-25M-shaped, k=%d)" % k,
        "value": round(args.steps / elapsed, 3),
        "unit": "epochs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": data_desc,
        "config": {
            "workload": "MovieLens-25M%s, als-implicit k=%d, %d timed epochs, %d x MI355X"
            % ("" if data_source != "synthetic" else "-shaped", k, args.steps, world),
            "data_source": data_source,
            "solver": "cholesky" if eng.u_plan.solver == 0 else "cg",
            "n_users": info["n_users"],
            "n_items": info["n_items"],
            "nnz": info["nnz"],
            "reg": reg,
            "weight": weight,
            "parallelism": ("row-sharded x%d, %d row slices per half-epoch (gathers under the solve)"
                            % (world, getattr(eng, "slices", 1))) if world > 1 else "single GPU",
        },
        "final_deltas": list(deltas),
        "setup_seconds": round(setup_seconds, 4),
    }
    if dry and world > 1:
        out["dryrun"] = True  # ranks shared ONE GPU over gloo: a rehearsal of the flow, not a measurement
    if roof:
        out["roofline"] = roof
    if getattr(eng, "summation_order", None):
        out["summation_order"] = eng.summation_order
    if getattr(eng, "collectives", None):
        out["collectives"] = eng.collectives

    # The secondary legs must never cost the headline line: a failure is reported in place.
    def leg(name, fn):
        try:
            out[name] = fn()
        except Exception as exc:  # noqa: BLE001 -- reported, not swallowed
            out[name] = {"error": f"{type(exc).__name__}: {exc}"}

    def als_parity_leg():
        frac = 1.0 if k <= 64 else max(0.02, (64.0 / k) ** 3)
        par, cpu = als_parity_and_cpu(eng, ui, k, reg, frac)
        out["cpu_baseline"] = cpu
        return par

    def topk_leg():
        # dense scoring + top-100 for ALL users with the factors just trained (north star:
        # "batched dense top-K scoring", f32 MFMA); exclusion of the users' own items included
        from lkpy_amd import _device as D

        h_ptr = eng.u_plan.csr.h_indptr.astype(np.int64)
        excl_ptr = torch.from_numpy(h_ptr).to(dev)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            g_idx, g_sc = D.score_topk(eng.P, eng.Q, k, 100, excl_ptr, eng.u_plan.csr.indices)
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t0)
        tb = min(ts)
        B, I = eng.P.shape[0], eng.Q.shape[0]
        fl = 2.0 * B * I * k
        res = {
            "metric": "dense scoring + top-100 of all users x all items (k=%d), seconds" % k,
            "value": round(tb, 4),
            "unit": "s",
            "users_per_s": round(B / tb, 1),
            "roofline": {
                "bound": "mfma", "achieved": round(fl / tb / 1e12, 2),
                "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(fl / tb / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                "algorithmic_flops": fl,
                "algorithmic_bytes": B * k * 4 + I * k * 4 * ((B + 2047) // 2048) + B * 100 * 8,
                "note": "end to end: GEMM + exclusion + selection, whole call wall time",
            },
        }
        tr, src = pmc_traffic_topk()
        res["roofline"]["traffic"] = tr
        res["roofline"]["traffic_source"] = (
            f"profiles/{src}: sum over the call's kernels of (2*FETCH_SIZE + WRITE_SIZE)*1024"
            if src else None)
        if not args.no_cpu:
            try:
                res["cpu_baseline"], res["parity"] = topk_cpu_and_parity(
                    backend.download(eng.P), backend.download(eng.Q), h_ptr,
                    eng.u_plan.csr.indices.cpu().numpy(), g_idx.cpu().numpy(),
                    g_sc.cpu().numpy(), 100)
            except Exception as exc:  # noqa: BLE001
                res["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"}
        return res

    def knn_leg():
        from lkpy_amd import _knn_bench

        res = _knn_bench.run(ratings, dev, checker=None if args.no_cpu else knn_cpu_and_parity,
                             score_checker=None if args.no_cpu else knn_score_cpu_and_parity,
                             recommend_checker=None if args.no_cpu else
                             knn_recommend_cpu_and_parity)
        if isinstance(res.get("recommend"), dict) and "error" not in res["recommend"]:
            try:
                res["recommend"]["through_components"] = knn_component_recommend(ratings)
            except Exception as exc:  # noqa: BLE001 -- reported in place
                res["recommend"]["through_components"] = {"error": f"{type(exc).__name__}: {exc}"}
        if isinstance(res.get("roofline"), dict) and args.scale == 1.0:
            # HBM bytes of the build kernel from the committed PMC summary of the same workload
            res["roofline"]["traffic"], res["roofline"]["traffic_source"] = pmc_traffic(
                ("r*_knn_counters.csv", "r*_knn_sym_counters.csv"), "iknn_build_kernel")
        return res

    def sharded_legs():
        """
        N > 1: the two legs that shard without a data-path collective (lkpy_amd/_sharded.py):
        every rank scores its block of users / builds its block of similarity rows; the time
        is barrier-bracketed and the maximum over ranks.  Blocks stay on their ranks.
        """
        from lkpy_amd import _device as D
        from lkpy_amd import _sharded

        def timed(fn):
            barrier()
            t0 = time.perf_counter()
            fn()
            barrier()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        res = {}
        if not args.no_topk:
            hp = eng.u_plan.csr.full_h_indptr
            if hp is None:  # LK_ALS_SETUP=sharded: no rank holds every user's row
                from lkpy_amd._als_engine import relabelled_user_lists

                hp, ex = relabelled_user_lists(ratings, eng.u_old, eng.i_new)
                excl_idx = torch.from_numpy(ex).to(dev)
            else:
                excl_idx = eng.u_plan.csr.indices
            excl_ptr = torch.from_numpy(hp.astype(np.int64)).to(dev)
            run = lambda: _sharded.score_topk_sharded(  # noqa: E731
                eng.P, eng.Q, k, 100, excl_ptr, excl_idx, collect=False)
            run()
            tb = timed(run)
            fl = 2.0 * eng.P.shape[0] * eng.Q.shape[0] * k
            res["topk"] = {"metric": "dense scoring + top-100 of all users x all items, users "
                           "sharded over %d GPUs, seconds" % world, "value": round(tb, 4),
                           "unit": "s", "users_per_s": round(eng.P.shape[0] / tb, 1),
                           "mfma_frac_aggregate": round(fl / tb / 1e12 / F32_MFMA_PEAK_TFLOPS
                                                        / world, 4)}
        if not args.no_knn:
            dui, diu, _means, _ = D.iknn_prepare(ratings, True, dev)
            run = lambda: _sharded.iknn_build_sharded(dui, diu, 1.0e-6, None, collect=False)  # noqa: E731
            run()
            tb = timed(run)
            res["knn"] = {"metric": "item-kNN model build seconds, output rows sharded over %d "
                          "GPUs (blocks left on their ranks)" % world, "value": round(tb, 4),
                          "unit": "s", "higher_is_better": False}
        return res

    if world > 1 and args.sharded_legs:
        # The headline is measured; these two legs have never run on more than one GPU.  A rank
        # that fails alone would leave the others in a collective for ever and the run without
        # its line: every rank arms the same watchdog, rank 0's prints the headline line as it
        # stands, then all leave.
        import threading

        def bail():
            if rank == 0:
                out["sharded_legs_error"] = ("timeout: no answer from the sharded legs in "
                                             "%d s, headline line emitted by the watchdog"
                                             % args.sharded_legs_timeout)
                out["incomplete"] = True  # (a hung leg must not read as success)
                emit(out)
                sys.stdout.flush()
            os._exit(3)  # the line is out, marked incomplete -- and the run does not read as a success

        dog = threading.Timer(args.sharded_legs_timeout, bail)
        dog.daemon = True
        dog.start()
        try:
            extra = sharded_legs()
        except Exception as exc:  # noqa: BLE001 -- reported, not swallowed
            extra = {"sharded_legs_error": f"{type(exc).__name__}: {exc}"}
        dog.cancel()
        out.update(extra)

    def cg_leg():
        """the conjugate-gradient row solve ``north_star`` names (the reference
        itself solves exactly: SURVEY section 0).  ``solver = "cg"`` from the trained state: epoch
        time beside the exact solver's, and how far one CG epoch lands from the exact epoch run
        from the same factors."""
        Ph, Qh = eng.user_embeddings(), eng.item_embeddings()
        out_cg = {}
        engs = {}
        for name, solver in (("exact", _native.SOLVER_CHOLESKY), ("cg", _native.SOLVER_CG)):
            e2 = ImplicitALSEngine(ui, k, reg, reg, Ph, Qh, HipBackend(k, dev, solver))
            if name == "cg":
                e2.u_plan.set_cg(CG_TOL, 0)
                e2.i_plan.set_cg(CG_TOL, 0)
            e2.train_epoch()
            e2.check()
            engs[name] = (e2.user_embeddings(), e2.item_embeddings())
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(3):
                e2.train_epoch()
            torch.cuda.synchronize(dev)
            out_cg[name + "_ms_per_epoch"] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
            e2.check()
            if name == "cg":
                (iu_, ru_), (ii_, ri_) = e2.u_plan.cg_stats(), e2.i_plan.cg_stats()
                out_cg["cg_rows"] = {"user": ru_, "item": ri_,
                                     "of": [int(np.sum(np.diff(ui.indptr) > 0)),
                                            int(np.sum(np.bincount(ui.indices,
                                                                   minlength=ui.shape[1]) > 0))]}
                out_cg["cg_iterations_per_row"] = {"user": round(iu_ / max(ru_, 1), 2),
                                                   "item": round(ii_ / max(ri_, 1), 2)}
            del e2
        rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))  # noqa: E731

        def rows(a, b):
            nb = np.linalg.norm(b, axis=1)
            d = np.linalg.norm(a - b, axis=1) / np.maximum(nb, 1e-30)
            d[nb == 0] = 0.0
            return {"row_rel_max": float(d.max()), "rows_over_1e-4": int((d > 1e-4).sum())}

        # (two instances: a workgroup per row, a wave per row; one launch each per half-epoch)
        kp = 64 if k <= 64 else (128 if k <= 128 else 256)
        tr4, src = pmc_traffic("r*_cg_k%d_counters.csv" % k, "als_cg_kernel<%d, false, 4>" % kp)
        tr1, _ = pmc_traffic("r*_cg_k%d_counters.csv" % k, "als_cg_kernel<%d, false, 1>" % kp)
        tr = None if tr4 is None or tr1 is None else tr4 + tr1
        return {"what": "solver='cg' vs the exact solver, one epoch from the same trained "
                "factors: Jacobi-preconditioned matrix-free CG (tol %.1e, warm start) on the rows " % CG_TOL
                +
                "whose gathered factor rows stay in registers over the iterations (<= 16384 / k' "
                "entries), the exact kernels on the longer rows (csrc/als_cg.hip); `user_rows` compares the two "
                "solvers from IDENTICAL inputs, `item_rows` from each engine's own user half (the "
                "difference of the inputs, up to `user_rows.row_rel_max` per row, is in it)",
                **out_cg,
                "one_epoch_rel_diff_P": rel(engs["cg"][0], engs["exact"][0]),
                "one_epoch_rel_diff_Q": rel(engs["cg"][1], engs["exact"][1]),
                "user_rows": rows(engs["cg"][0], engs["exact"][0]),
                "item_rows": rows(engs["cg"][1], engs["exact"][1]),
                "hbm_bytes_per_half_epoch_cg_kernels_from_pmc": tr, "pmc_source": src}

    def k128_leg():
        """BASELINE.json configs[3] on ONE GPU: the same data at k = 128 (``als_blk.hip``): a few
        timed epochs, roofline, and the parity / cpu_baseline pair on a row sample."""
        k2 = 128
        rng2 = np.random.default_rng(42)
        Q2 = rng2.standard_normal((ui.shape[1], k2), dtype=np.float32) * 0.01
        Q2 *= Q2
        P2 = rng2.standard_normal((ui.shape[0], k2), dtype=np.float32) * 0.01
        P2 *= P2
        eng2, _b2, el2, roof2, setup2, deltas2 = als_timed(
            ui, P2, Q2, k2, reg, args.k128_steps, 3, dev, world, args.scale)
        res = {
            "metric": "ALS-implicit epochs/sec (ML-25M-shaped, k=128), 1 x MI355X",
            "value": round(args.k128_steps / el2, 3), "unit": "epochs/s",
            "steps": args.k128_steps, "warmup": 3,
            "ms_per_step": round(el2 / args.k128_steps * 1e3, 4),
            "setup_seconds": round(setup2, 4), "final_deltas": list(deltas2),
            "config": {"workload": "MovieLens-25M-shaped, als-implicit k=128 (BASELINE configs[3] "
                       "on one GPU), %d timed epochs" % args.k128_steps},
            "roofline": roof2,
        }
        if getattr(eng2, "summation_order", None):
            res["summation_order"] = eng2.summation_order
        if not args.no_cpu:
            res["parity"], res["cpu_baseline"] = als_parity_and_cpu(eng2, ui, k2, reg, 0.05)
        return res

    def cfg5_leg():
        "BASELINE.json configs[4] on ONE GPU: 3 timed epochs + parity sample + a top-K slice"
        return cfg5_run(args, dev, world, rank, 3, 1, args.topk_users)

    single = rank == 0 and world == 1
    if single and not args.no_cpu:
        leg("parity", als_parity_leg)
    if single and not args.no_topk:
        leg("topk", topk_leg)
    if single and not args.no_cg and k >= 64:
        leg("cg", cg_leg)
    if single:
        del eng  # free the engine's HBM before the other legs
        torch.cuda.empty_cache()
    if single and not args.no_fit:
        kept = {}
        leg("fit", lambda: fit_leg(ratings, k, 20, weight, kept))  # configs[1]: 20 epochs
        if kept and not args.no_recommend:
            leg("recommend", lambda: recommend_leg(kept["scorer"], kept["ds"], 10000, 100,
                                                   args.no_cpu))
        kept.clear()
        torch.cuda.empty_cache()
    if single and not args.no_knn:
        leg("knn", knn_leg)
        if isinstance(out.get("parity"), dict) and isinstance(out["knn"], dict):
            out["parity"]["knn"] = out["knn"].pop("parity", None)
    if single and not args.no_k128 and k == 64:
        leg("k128", k128_leg)
        torch.cuda.empty_cache()
    if single and not args.no_cfg5 and k == 64:
        leg("cfg5", cfg5_leg)
    if rank == 0:
        emit(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
