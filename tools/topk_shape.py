#!/usr/bin/env python3
"""
Fused top-N at an arbitrary shape, round-4 against round-5 kernels in one process:
    python tools/topk_shape.py B I k [n] [excl_per_user]
random factors with popularity-skewed item norms, `excl_per_user` random exclusions per row.
One JSON line per variant (knobs as in tools/topk_ab.py; lists compared with the first variant).
"""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lkpy_amd import _device as D  # noqa: E402

B, I, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 100
ne = int(sys.argv[5]) if len(sys.argv) > 5 else 10
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(7)
kp = D.padded_dim(k)
P = torch.zeros(B, kp, device=dev)
Q = torch.zeros(I, kp, device=dev)
P[:, :k] = torch.randn(B, k, device=dev, generator=g) * 0.1
Q[:, :k] = torch.randn(I, k, device=dev, generator=g) * 0.1 * (
    0.2 + torch.rand(I, 1, device=dev, generator=g))
excl_ptr = torch.arange(0, (B + 1) * ne, ne, device=dev, dtype=torch.int64)
excl_idx = torch.randint(0, I, (B * ne,), device=dev, generator=g, dtype=torch.int32)
variants = [x for x in os.environ.get(
    "LK_SHAPE_VARIANTS",
    "LK_TOPK_STAGE1=panel,LK_TOPK_SELECT=sort,LK_TOPK_SAMPLE_DIV=16;"
    "LK_TOPK_STAGE1=cmax,LK_TOPK_SELECT=wave,LK_TOPK_SAMPLE_DIV=24;"
    "LK_TOPK_STAGE1=cmax,LK_TOPK_SELECT=wave,LK_TOPK_SAMPLE_DIV=16;"
    "LK_TOPK_STAGE1=panel,LK_TOPK_SELECT=wave,LK_TOPK_SAMPLE_DIV=16;"
    "LK_TOPK_STAGE1=cmax,LK_TOPK_SELECT=sort,LK_TOPK_SAMPLE_DIV=16").split(";") if x]
ref = None
for setting in variants:
    pairs = [kv.split("=") for kv in setting.split(",")]
    for kk, vv in pairs:
        os.environ[kk] = vv
    ts = []
    for _ in range(int(os.environ.get("LK_SHAPE_REPS", "3"))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx, sc = D.score_topk(P, Q, k, n, excl_ptr, excl_idx)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    rec = {"knobs": setting, "ms": round(min(ts) * 1e3, 2), "all_ms": [round(t * 1e3, 1) for t in ts],
           "tflops": round(2.0 * B * I * k / min(ts) / 1e12, 1)}
    if ref is None:
        ref = idx.clone()
    else:
        rec["lists_identical"] = bool(torch.equal(idx, ref))
    print(json.dumps(rec), flush=True)
    for kk, _ in pairs:
        os.environ.pop(kk, None)
