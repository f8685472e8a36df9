#!/usr/bin/env python3
"Dense scoring + top-N timing only (ML-25M-shaped sizes, random factors): python tools/topk_only.py [k] [n]"
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lkpy_amd import _device as D  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
B, I = 162541, 62423
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(7)
kp = D.padded_dim(k)
P = torch.zeros(B, kp, device=dev)
Q = torch.zeros(I, kp, device=dev)
P[:, :k] = torch.randn(B, k, device=dev, generator=g) * 0.1
Q[:, :k] = torch.randn(I, k, device=dev, generator=g) * 0.1
times = []
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    idx, sc = D.score_topk(P, Q, k, n, None, None)
    torch.cuda.synchronize()
    times.append(time.perf_counter() - t0)
dt = min(times)
print(json.dumps({"users": B, "items": I, "k": k, "n": n, "seconds": round(dt, 4),
                  "all": [round(t, 4) for t in times], "users_per_s": round(B / dt, 1),
                  "tflops": round(2.0 * B * I * k / dt / 1e12, 2),
                  "check": int(idx[:5, :3].sum().item())}))
