#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the MI355X backend (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Workload (N = 1): BASELINE.json configs[1] -- "MovieLens-25M, als-implicit k=64, 20
epochs, 1 x MI355X".  MovieLens-25M itself is not on the box (no network), so the
input is the seeded ML-25M-shaped synthetic of ``lkpy_amd.synth`` (same user/item
counts, ~same nnz and activity skew).  A *step* is ONE ALS
EPOCH: user half-epoch + item half-epoch + both Gramians (+ the exchanges when
N > 1), with CSR and factors already resident in HBM.  ``value`` = epochs/second.

For N > 1 the driver launches one process per GPU (torch.distributed.run); users and
items are row-sharded (lkpy_amd._als_engine) and total work is fixed => "strong".

One JSON line on rank 0.  Extra objects:
  roofline     -- the dominant kernel (als_solve_kernel, f32 MFMA bound): algorithmic
                  flops per launch / average launch duration (HIP events recorded on
                  the launch stream inside the library, lk_als_plan_get_timing).
  cpu_baseline -- the CPU oracle (a port of the reference's Rust + LAPACK path) timed
                  on this box's host cores on a bounded row sample (rank 0, N = 1).
  knn          -- item-kNN model build seconds on the same data (BASELINE.json
                  metric's second half), when the build kernel is available.
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

F32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: peak FP32 (matrix)
LONG_ROW = int(os.environ.get("LK_BENCH_LONG_ROW", 2048))  # LK_ALS_LONG_ROW in lkpy_amd/csrc/als_chol.hip


def half_flops(lengths: np.ndarray, k: int):
    """
    Algorithmic flops of one half-epoch:
    nnz*(2k^2 + 2k) + rows_nonempty*(k^3/3 + 2k^2), split into the part done by the
    solve kernel (short rows + every solve) and by the chunk kernel (long rows' Gram).
    """
    lengths = lengths.astype(np.int64)
    per_nnz = 2 * k * k + 2 * k
    per_row = k**3 / 3.0 + 2 * k * k
    long_nnz = int(lengths[lengths > LONG_ROW].sum())
    short_nnz = int(lengths.sum()) - long_nnz
    nonempty = int((lengths > 0).sum())
    return short_nnz * per_nnz + nonempty * per_row, long_nnz * per_nnz


def half_bytes(lengths: np.ndarray, k: int):
    "Algorithmic HBM bytes of one half-epoch (SURVEY.md section 8d)."
    nnz, rows = int(lengths.sum()), len(lengths)
    return nnz * (4 + 4 + 4 * k) + (rows + 1) * 4 + rows * k * 4 * 2 + k * k * 4


def pmc_traffic(kernel_substr: str):
    """
    HBM bytes per launch of a kernel from the COMMITTED rocprofv3 PMC summary
    (profiles/*_counters.csv, written by tools/prof_als.sh + tools/summarize_prof.py on the
    same workload): (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- FETCH_SIZE is doubled because on
    gfx950 it reports half the bytes of wide (16 B/lane) coalesced reads
    (MI355X_MICROARCH.md, section HBM).  Counters cannot be collected inside this process, so
    the value is null when no summary is committed.
    """
    import csv

    files = sorted((ROOT / "profiles").glob("r*_als_*_counters.csv"))
    if not files:
        return None, None
    fetch, write = [], []
    with open(files[-1]) as f:
        for row in csv.DictReader(f):
            if kernel_substr in row["Kernel_Name"]:
                if row["Counter_Name"] == "FETCH_SIZE":
                    fetch.append(float(row["mean"]))
                elif row["Counter_Name"] == "WRITE_SIZE":
                    write.append(float(row["mean"]))
    if not fetch:
        return None, None
    w = sum(write) / len(write) if write else 0.0
    return (2.0 * sum(fetch) / len(fetch) + w) * 1024.0, files[-1].name


def cpu_baseline(ui, k, reg, budget_s=20.0):
    """
    Time the CPU oracle (port of src/accel/als/implicit.rs + LAPACK sposv) on a row
    sample of the same workload and extrapolate to epochs/second by nnz.
    """
    from oracle import lk_oracle as lko

    import scipy.sparse as sps

    rng = np.random.default_rng(1)
    iu = sps.csr_array(ui.T)
    iu.sort_indices()
    P = lko.als_initial_params(rng, ui.shape[0], k)
    Q = lko.als_initial_params(rng, ui.shape[1], k)
    threads = lko.num_threads()
    est = 0.0
    used = 0.0
    desc = []
    for name, mat, this, other in (("user", ui, P, Q), ("item", iu, Q, P)):
        n = mat.shape[0]
        # calibrate on 0.5 % of the rows, then take what fits the budget
        frac = 0.005
        for _ in range(2):
            rows = np.sort(rng.choice(n, max(64, int(n * frac)), replace=False))
            sub = sps.csr_array(mat[rows])
            tt = np.ascontiguousarray(this[rows])
            otor = lko.implicit_otor(other, reg)
            t0 = time.perf_counter()
            lko.als_half_epoch(sub, tt, other, otor, threads)
            dt = time.perf_counter() - t0
            used += dt
            full = dt * mat.nnz / max(sub.nnz, 1)
            frac = min(1.0, frac * (budget_s / 2) / max(dt, 1e-3) * 0.5)
        est += full
        desc.append(f"{name} half: {len(rows)} of {n} rows ({sub.nnz} nnz) in {dt:.2f}s")
    return {
        "value": 1.0 / est,
        "unit": "epochs/s",
        "cores": threads,
        "kind": "port",
        "sample": "; ".join(desc) + "; extrapolated by nnz to a full epoch",
        "host_cpus": os.cpu_count(),
        "cpu_seconds_used": round(used, 2),
    }


def cpu_baseline_knn(ratings, budget_s=15.0):
    """
    Time the CPU oracle's ``sim_row`` (port of src/accel/knn/item_train.rs:95-152) on a
    random sample of item rows and extrapolate the full build by multiply-accumulates.
    """
    from lkpy_amd._knn_bench import prepare_explicit
    from oracle import lk_oracle as lko

    ui, iu, _ = prepare_explicit(ratings)
    ulen = np.diff(ui.indptr).astype(np.int64)
    row_macs = np.add.reduceat(ulen[iu.indices], iu.indptr[:-1].astype(np.int64))
    row_macs[np.diff(iu.indptr) == 0] = 0
    total = int(row_macs.sum())
    rng = np.random.default_rng(2)
    threads = min(lko.num_threads(), os.cpu_count() or 1)
    n = max(64, ui.shape[1] // 400)
    dt, rows = 0.0, None
    for _ in range(3):
        rows = np.sort(rng.choice(ui.shape[1], min(n, ui.shape[1]), replace=False))
        t0 = time.perf_counter()
        lko.iknn_sample_rows(ui, iu, rows, 1.0e-6, None, threads)
        dt = time.perf_counter() - t0
        if dt > budget_s / 4 or n >= ui.shape[1]:
            break
        n = int(min(ui.shape[1], n * min(8.0, budget_s / 2 / max(dt, 1e-3))))
    frac = float(row_macs[rows].sum()) / max(total, 1)
    return {
        "value": round(dt / max(frac, 1e-12), 2),
        "unit": "s",
        "cores": threads,
        "kind": "port",
        "sample": f"{len(rows)} of {ui.shape[1]} item rows ({frac * 100:.2f}% of the "
        f"multiply-accumulates) in {dt:.2f}s, extrapolated by MACs",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--k", type=int, default=64)
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the dataset (debug only)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-knn", action="store_true", help="skip the item-kNN build leg")
    ap.add_argument("--no-topk", action="store_true", help="skip the dense top-N scoring leg")
    args = ap.parse_args()

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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    k, reg, weight = args.k, 0.1, 40.0
    ratings = synth.ml25m_like(scale=args.scale)
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

    backend = HipBackend(k, dev, _native.SOLVER_AUTO)
    eng = ImplicitALSEngine(ui, k, reg, reg, P0, Q0, backend)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        eng.train_epoch()
    eng.check()
    timing_ok = hasattr(eng.u_plan, "enable_timing")
    if timing_ok:
        eng.u_plan.enable_timing(True)
        eng.i_plan.enable_timing(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        du, di = eng.train_epoch()
    barrier()
    elapsed = time.perf_counter() - t0
    eng.check()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3

    # ---- roofline of the dominant kernel (local shard of this rank) ----
    roof = None
    if timing_ok:
        cu, su, nu = eng.u_plan.get_timing()
        ci, si, ni = eng.i_plan.get_timing()
        ulen = np.diff(eng.u_plan.csr.h_indptr)
        ilen = np.diff(eng.i_plan.csr.h_indptr)
        fu_solve, fu_chunk = half_flops(ulen, k)
        fi_solve, fi_chunk = half_flops(ilen, k)
        launches = nu + ni
        if launches > 0 and (su + si) > 0:
            flops_per_launch = (fu_solve * nu + fi_solve * ni) / launches
            avg_ms = (su + si) / launches
            achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
            roof = {
                "kernel": "als_solve_kernel<NT=%d>" % (backend.kp // 16),
                "bound": "mfma",
                "achieved": round(achieved, 3),
                "peak": F32_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 4),
                "traffic": None,
                "traffic_source": None,
                "avg_launch_ms": round(avg_ms, 4),
                "launches": launches,
                "algorithmic_flops_per_launch": flops_per_launch,
                "algorithmic_bytes_per_launch": (half_bytes(ulen, k) * nu + half_bytes(ilen, k) * ni)
                / launches,
                "chunk_kernel_ms_per_launch": round((cu + ci) / launches, 4),
                "chunk_kernel_flops_per_launch": (fu_chunk * nu + fi_chunk * ni) / launches,
            }

    if roof and world == 1 and args.scale == 1.0 and k == 64:
        roof["traffic"], roof["traffic_source"] = pmc_traffic("als_solve_kernel")
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
        "data": "synthetic (seeded ML-25M-shaped: lkpy_amd.synth.ml25m_like, seed 20260925)",
        "config": {
            "workload": "MovieLens-25M-shaped, als-implicit k=%d, %d timed epochs, %d x MI355X"
            % (k, args.steps, world),
            "solver": "cholesky" if eng.u_plan.solver == 0 else "cg",
            "n_users": info["n_users"],
            "n_items": info["n_items"],
            "nnz": info["nnz"],
            "reg": reg,
            "weight": weight,
            "parallelism": "row-sharded x%d" % world if world > 1 else "single GPU",
        },
        "final_deltas": [float(du.item()), float(di.item())],
    }
    if roof:
        out["roofline"] = roof

    # The secondary legs must never cost the headline line: a failure is reported in place.
    def leg(name, fn):
        try:
            out[name] = fn()
        except Exception as exc:  # noqa: BLE001 -- reported, not swallowed
            out[name] = {"error": f"{type(exc).__name__}: {exc}"}

    def topk_leg():
        # dense scoring + top-100 for ALL users with the factors just trained (north star:
        # "batched dense top-K scoring", f32 MFMA); exclusion of the users' own items included
        from lkpy_amd import _device as D

        excl_ptr = torch.from_numpy(eng.u_plan.csr.h_indptr.astype(np.int64)).to(dev)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            D.score_topk(eng.P, eng.Q, k, 100, excl_ptr, eng.u_plan.csr.indices)
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t0)
        tb = min(ts)
        fl = 2.0 * eng.P.shape[0] * eng.Q.shape[0] * k
        return {
            "metric": "dense scoring + top-100 of all users x all items (k=%d), seconds" % k,
            "value": round(tb, 4),
            "unit": "s",
            "users_per_s": round(eng.P.shape[0] / tb, 1),
            "achieved_tflops": round(fl / tb / 1e12, 2),
            "mfma_frac_end_to_end": round(fl / tb / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
            "note": "GEMM + exclusion mask + selection; the score panel kernel alone reaches "
            "0.62 of the f32 MFMA peak at k=128 (profiles/r01_topk_*)",
        }

    def knn_leg():
        from lkpy_amd import _knn_bench

        res = _knn_bench.run(ratings, dev)
        if not args.no_cpu:
            try:
                res["cpu_baseline"] = cpu_baseline_knn(ratings)
            except Exception as exc:  # noqa: BLE001
                res["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"}
        return res

    if rank == 0 and world == 1 and not args.no_topk:
        leg("topk", topk_leg)
    if rank == 0 and world == 1 and not args.no_knn:
        leg("knn", knn_leg)
    if rank == 0 and world == 1 and not args.no_cpu:
        leg("cpu_baseline", lambda: cpu_baseline(ui, k, reg))
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
