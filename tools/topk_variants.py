#!/usr/bin/env python3
"""
Time dense scoring + top-N with several builds of the library in ONE process:
    python tools/topk_variants.py [k] [n] tools/_variants/lkamd_*.so
(ML-25M-shaped sizes, random factors with popularity-skewed item norms, no exclusions).
"""
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lkpy_amd import _native  # noqa: E402

k, n = int(sys.argv[1]), int(sys.argv[2])
B, I = 162541, 62423
dev = torch.device("cuda:0")
default = _native.LIB_PATH
for path in [default] + [Path(p).resolve() for p in sys.argv[3:]]:
    _native._lib = None
    _native.LIB_PATH = Path(path)
    from lkpy_amd import _device as D

    g = torch.Generator(device=dev).manual_seed(7)
    kp = D.padded_dim(k)
    P = torch.zeros(B, kp, device=dev)
    Q = torch.zeros(I, kp, device=dev)
    P[:, :k] = torch.randn(B, k, device=dev, generator=g) * 0.1
    Q[:, :k] = torch.randn(I, k, device=dev, generator=g) * 0.1
    times = []
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx, sc = D.score_topk(P, Q, k, n, None, None)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    dt = min(times)
    print(json.dumps({"lib": Path(path).name, "ms": round(dt * 1e3, 2),
                      "tflops": round(2.0 * B * I * k / dt / 1e12, 1),
                      "check": int(idx[:5, :3].sum().item())}), flush=True)
