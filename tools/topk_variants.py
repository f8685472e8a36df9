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
    for _ in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx, sc = D.score_topk(P, Q, k, n, None, None)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    dt = min(times)
    w = torch.arange(1, n + 1, device=dev, dtype=torch.int64)
    rec = {"lib": Path(path).name, "ms": round(dt * 1e3, 2),
           "tflops": round(2.0 * B * I * k / dt / 1e12, 1),
           "check": int(((idx.long() * w).sum(1) % 1000003).sum().item())}  # same lists, all rows
    lib = _native.load()
    if hasattr(lib, "lk_topk_phase_set"):  # -DLK_TOPK_PHASES build: cycles per phase and wave
        import ctypes

        import numpy as np

        buf = torch.zeros((512 * 8, 8), dtype=torch.int64, device=dev)
        lib.lk_topk_phase_set(ctypes.c_void_p(buf.data_ptr()))
        D.score_topk(P, Q, k, n, None, None)
        torch.cuda.synchronize()
        lib.lk_topk_phase_set(ctypes.c_void_p(0))
        b = buf.cpu().numpy()
        names = ["wait+stage", "mfma", "bar_loop", "compare", "flush", "bar_epi", "tiles", "whole"]
        nw = 8 if (b[512 * 4:, 7] > 0).any() else 4  # waves per workgroup of this build
        b = b[: 512 * nw]
        last = (B - 2 * 65536 + 127) // 128  # workgroups of the last batch (one per CU)
        for tag, x in [("one_wg_per_cu", b[: last * nw]), ("two_wg_per_cu", b[last * nw:])]:
            rec[tag] = {nm: int(np.mean(x[:, i])) for i, nm in enumerate(names)}
            hw = (x[:, 6] >> 32) & 0xffffffff
            if hw.any():  # stagger build: HW_ID of the wave
                rec[tag]["wave_id"] = np.bincount(hw & 15, minlength=16).tolist()
                rec[tag]["tg_id"] = np.bincount((hw >> 16) & 15, minlength=16).tolist()
            rec[tag]["whole_pct"] = [int(v) for v in np.percentile(x[:, 7], [0, 10, 50, 90, 100])]
    print(json.dumps(rec), flush=True)
