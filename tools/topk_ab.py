#!/usr/bin/env python3
"""
A/B of the fused top-N path's stage 1 and stage 3 in ONE process (the knobs are read per call):
    python tools/topk_ab.py [k] [n] [epochs]
cfg2-shaped synthetic ratings, factors from a few implicit-ALS epochs, every user's own items
excluded (the BENCH workload).  Variants: LK_TOPK_STAGE1 = cmax | panel  x  LK_TOPK_SELECT = wave | sort;
the round-4 pair (panel, sort) is the reference the other lists and score bits are compared with.
Prints one JSON line per variant (per-kernel times: run it under rocprofv3 --kernel-trace).
"""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import scipy.sparse as sps
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lkpy_amd import _device as D, synth  # noqa: E402
from lkpy_amd._als_engine import HipBackend, ImplicitALSEngine  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 4
scale = float(os.environ.get("LK_AB_SCALE", "1.0"))
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
ratings = synth.ml25m_like(seed=3, scale=scale)
ui = sps.csr_array((np.full(ratings.nnz, 40.0, np.float32), ratings.indices, ratings.indptr),
                   shape=ratings.shape)
rng = np.random.default_rng(0)
P0 = (rng.standard_normal((ui.shape[0], k)) * 0.01).astype(np.float32)
Q0 = (rng.standard_normal((ui.shape[1], k)) * 0.01).astype(np.float32)
eng = ImplicitALSEngine(ui, k, 0.1, 0.1, P0, Q0, HipBackend(k, dev))
for _ in range(epochs):
    eng.train_epoch()
excl_ptr = torch.from_numpy(eng.u_plan.csr.h_indptr.astype(np.int64)).to(dev)
excl_idx = eng.u_plan.csr.indices
B, I = eng.P.shape[0], eng.Q.shape[0]
ref = None
for s1, s3 in (("panel", "sort"), ("cmax", "sort"), ("panel", "wave"), ("cmax", "wave")):
    os.environ["LK_TOPK_STAGE1"] = s1
    os.environ["LK_TOPK_SELECT"] = s3
    ts = []
    for _ in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx, sc = D.score_topk(eng.P, eng.Q, k, n, excl_ptr, excl_idx)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    rec = {"stage1": s1, "select": s3, "ms": round(min(ts) * 1e3, 3),
           "all_ms": [round(t * 1e3, 2) for t in ts],
           "tflops": round(2.0 * B * I * k / min(ts) / 1e12, 1)}
    if ref is None:
        ref = (idx.clone(), sc.clone())
    else:
        rec["lists_identical"] = bool(torch.equal(idx, ref[0]))
        rec["score_bits_identical"] = bool(torch.equal(sc.view(torch.int32), ref[1].view(torch.int32)))
        if not rec["lists_identical"]:
            bad = (idx != ref[0]).any(1).nonzero().flatten()
            rec["rows_differing"] = int(bad.numel())
            rec["first_bad_rows"] = bad[:5].tolist()
    print(json.dumps(rec), flush=True)
os.environ.pop("LK_TOPK_STAGE1")
os.environ.pop("LK_TOPK_SELECT")
# knob sweeps on the default path (lists compared with the reference pair's):
# LK_AB_SWEEP="LK_TOPK_SCHEDULE=1;LK_TOPK_SAMPLE_DIV=16,LK_TOPK_SCHEDULE=1;..."
for setting in [x for x in os.environ.get("LK_AB_SWEEP", "").split(";") if x]:
    pairs = [kv.split("=") for kv in setting.split(",")]
    for kk, vv in pairs:
        os.environ[kk] = vv
    ts = []
    for _ in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx, sc = D.score_topk(eng.P, eng.Q, k, n, excl_ptr, excl_idx)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(json.dumps({"knobs": setting, "ms": round(min(ts) * 1e3, 3),
                      "median_ms": round(float(np.median(ts)) * 1e3, 3),
                      "lists_identical": bool(torch.equal(idx, ref[0]))}), flush=True)
    for kk, _ in pairs:
        os.environ.pop(kk, None)
# no exclusions at all (the scorer without a training-item mask)
ts = []
for _ in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    idx2, sc2 = D.score_topk(eng.P, eng.Q, k, n, None, None)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
os.environ["LK_TOPK_STAGE1"] = "panel"
os.environ["LK_TOPK_SELECT"] = "sort"
idx3, sc3 = D.score_topk(eng.P, eng.Q, k, n, None, None)
print(json.dumps({"no_exclusions_ms": round(min(ts) * 1e3, 3),
                  "lists_identical": bool(torch.equal(idx2, idx3)),
                  "score_bits_identical": bool(torch.equal(sc2.view(torch.int32),
                                                           sc3.view(torch.int32)))}), flush=True)

from lkpy_amd import _native  # noqa: E402

lib = _native.load()
if hasattr(lib, "lk_wsel_phase_set"):  # -DLK_WSEL_PHASES build (tools/build_variant.sh)
    import ctypes

    names = ["scalars", "cand loads", "hash insert", "excl walk", "look-up", "search", "compact",
             "sort", "output"]
    tn = ["row loads", "clear + walk", "keys to LDS", "repairs", "search + store"]
    os.environ["LK_TOPK_STAGE1"] = "cmax"
    os.environ["LK_TOPK_SELECT"] = "wave"
    rows_a = (((B + 127) // 128) // 512) * 512 * 128

    def table(x, cols):
        tot = x[:, 10] - x[:, 9]
        span = float(x[:, 10].max() - x[:, 9].min())
        i = int(np.argmax(tot))
        return {"rows": len(x), "mean_ticks": {nm: round(float(x[:, j].mean()), 1)
                                               for j, nm in enumerate(cols)},
                "mean_row": round(float(tot.mean()), 1), "p50": float(np.median(tot)),
                "p99": float(np.quantile(tot, 0.99)), "longest_row": float(tot[i]),
                "longest_row_aux": [int(x[i, 11]) & 0xffffffff, int(x[i, 11]) >> 32],
                "first_start_to_last_end": span,
                "mean_rows_in_flight": round(float(tot.sum()) / span, 1)}

    for label, ep, ei in (("exclusions", excl_ptr, excl_idx), ("none", None, None)):
        buf = torch.zeros((2, 262144, 16), dtype=torch.int64, device=dev)
        lib.lk_wsel_phase_set(ctypes.c_void_p(buf.data_ptr()))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        D.score_topk(eng.P, eng.Q, k, n, ep, ei)
        torch.cuda.synchronize()
        ms = round((time.perf_counter() - t0) * 1e3, 3)
        lib.lk_wsel_phase_set(ctypes.c_void_p(0))
        h = buf.cpu().numpy()
        sel, tau = h[0, :B], h[1, :B]
        done = sel[:, 10] > 0  # rows of the second tier leave no record
        print(json.dumps({"phases": label, "instrumented_call_ms": ms}), flush=True)
        for tag, x in (("select beside the filter", sel[:rows_a][done[:rows_a]]),
                       ("select alone", sel[rows_a:][done[rows_a:]])):
            print(json.dumps({"phases": label, "launch": tag, **table(x, names)}), flush=True)
        print(json.dumps({"phases": label, "launch": "cmax_tau", **table(tau, tn)}), flush=True)
