#!/usr/bin/env python3
"""
A/B of the ALS summation-order modes on the ML-25M shape (round 5): epoch time of
``LK_ALS_RHS_ORDER`` = auto (hybrid: rows > LK_ALS_REF_LEN entries in the reference's order) /
accurate (round 4's default) / reference (strict), and -- with --parity -- one epoch of each mode
from the SAME trained state against the CPU oracle from identical inputs, rows over 1e-4 listed
with their lengths.

    python tools/order_ab.py --k 64 --epochs 25 --parity [--ref-len 2048,1024]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=64)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--epochs", type=int, default=25, help="epochs to the trained state")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--parity", action="store_true")
    ap.add_argument("--modes", default="accurate,auto")
    ap.add_argument("--ref-len", default="", help="comma list of LK_ALS_REF_LEN values for auto")
    ap.add_argument("--row-frac", type=float, default=1.0)
    args = ap.parse_args()

    import scipy.sparse as sps
    import torch

    from lkpy_amd import _native, synth
    from lkpy_amd._als_engine import HipBackend, ImplicitALSEngine
    from oracle import lk_oracle as lko

    _native.require_gpu()
    dev = torch.device("cuda:0")
    k, reg = args.k, 0.1
    ratings = synth.ml25m_like(scale=args.scale)
    ui = sps.csr_array((np.full(ratings.nnz, 40.0, np.float32), ratings.indices, ratings.indptr),
                       shape=ratings.shape)
    rng = np.random.default_rng(0)
    Q0 = lko.als_initial_params(rng, ui.shape[1], k)
    P0 = lko.als_initial_params(rng, ui.shape[0], k)

    # a variant: mode[:ENV=VALUE[:ENV=VALUE ...]] (environment knobs the library reads per launch
    # or per plan, e.g. auto:LK_ALS_SIDE_STREAM=0)
    variants = []
    for m in args.modes.split(","):
        mode, *envs = m.split(":")
        if mode == "auto" and args.ref_len and not envs:
            variants += [("auto", [f"LK_ALS_REF_LEN={rl}"]) for rl in args.ref_len.split(",")]
        else:
            variants.append((mode, envs))

    # the trained state, once (accurate mode), shared by every variant
    eng0 = ImplicitALSEngine(ui, k, reg, reg, P0, Q0, HipBackend(k, dev, reference_order="accurate"))
    for _ in range(args.epochs):
        eng0.train_epoch()
    eng0.check()
    P, Q = eng0.user_embeddings(), eng0.item_embeddings()
    del eng0

    want = None
    if args.parity:
        iu = sps.csr_array(ui.T)
        iu.sort_indices()
        threads = lko.num_threads()
        rs = np.random.default_rng(5)

    out = {}
    touched = set()
    for mode, envs in variants:
        name = ":".join([mode] + envs)
        for key in touched:
            os.environ.pop(key, None)
        for e in envs:
            key, val = e.split("=", 1)
            os.environ[key] = val
            touched.add(key)
        eng = ImplicitALSEngine(ui, k, reg, reg, P, Q, HipBackend(k, dev, reference_order=mode))
        du, di = eng.train_epoch()
        eng.check()
        P1, Q1 = eng.user_embeddings(), eng.item_embeddings()
        res = {"long_rows": [eng.u_plan.long_rows(), eng.i_plan.long_rows()]}
        for _ in range(3):
            eng.train_epoch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.train_epoch()
        torch.cuda.synchronize()
        res["ms_per_epoch"] = (time.perf_counter() - t0) / args.steps * 1e3
        eng.u_plan.enable_timing(True)
        eng.i_plan.enable_timing(True)
        for _ in range(10):
            eng.train_epoch()
        torch.cuda.synchronize()
        cu, su, nu = eng.u_plan.get_timing()
        ci, si, ni = eng.i_plan.get_timing()
        res["user_chunk_ms"], res["user_solve_ms"] = cu / max(nu, 1), su / max(nu, 1)
        res["item_chunk_ms"], res["item_solve_ms"] = ci / max(ni, 1), si / max(ni, 1)
        if args.parity:
            par = {}
            for half, mat, this, other, got in (("user", ui, P, Q, P1), ("item", iu, Q, P1, Q1)):
                n = mat.shape[0]
                if args.row_frac < 1.0:
                    rows = np.sort(rs.choice(n, max(256, int(n * args.row_frac)), replace=False))
                    # (always the 64 longest rows)
                    lens_all = np.diff(mat.indptr)
                    rows = np.unique(np.concatenate([rows, np.argsort(-lens_all)[:64]]))
                    sub, t0_, got_ = sps.csr_array(mat[rows]), this[rows], got[rows]
                else:
                    sub, t0_, got_ = mat, this, got
                w = np.ascontiguousarray(t0_.copy())
                otor = lko.implicit_otor(other, reg)
                lko.als_half_epoch(sub, w, other, otor, threads)
                lens = np.diff(sub.indptr)
                nz = lens > 0
                num = np.linalg.norm(got_.astype(np.float64) - w, axis=1)
                den = np.maximum(np.linalg.norm(w.astype(np.float64), axis=1), 1e-300)
                rel = np.where(nz, num / den, 0.0)
                over = np.flatnonzero(rel > 1e-4)
                buckets = {}
                for lo, hi in ((0, 256), (256, 1024), (1024, 2048), (2048, 4096), (4096, 16384),
                               (16384, 1 << 30)):
                    m = (lens > lo) & (lens <= hi)
                    if m.any():
                        buckets[f"{lo}-{hi}"] = [int(m.sum()), float(rel[m].max()),
                                                 int((rel[m] > 1e-4).sum()),
                                                 int((rel[m] > 5e-5).sum())]
                par[half] = {"rows": int(nz.sum()), "over_1e-4": int(len(over)),
                             "rel_max": float(rel.max()),
                             "over_rows_len": [int(x) for x in lens[over][:20]],
                             "by_len(rows,max,over1e-4,over5e-5)": buckets}
            res["parity"] = par
        out[name] = res
        print(name, json.dumps(res), flush=True)
        del eng
    print(json.dumps(out))


if __name__ == "__main__":
    main()
