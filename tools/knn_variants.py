#!/usr/bin/env python3
"""
Item-kNN build (ML-25M-shaped, unbounded) with several builds of the library / window widths in
ONE process:  python tools/knn_variants.py [W=4096,2048 ...] tools/_variants/lkamd_*.so
Prints build seconds, the build kernel's milliseconds and a checksum of the whole CSR.
"""
import json
import os
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lkpy_amd import _native, synth  # noqa: E402

ratings = synth.ml25m_like()
dev = torch.device("cuda:0")
default = _native.LIB_PATH
widths = [a for a in sys.argv[1:] if a.isdigit()] or ["4096"]
libs = [default] + [Path(a).resolve() for a in sys.argv[1:] if not a.isdigit()]
for path in libs:
    for w in widths:
        os.environ["LK_IKNN_W"] = w
        _native._lib = None
        _native.LIB_PATH = Path(path)
        from lkpy_amd import _device as D

        dui, diu, _m, _ = D.iknn_prepare(ratings, True, dev)
        ts, ks = [], []
        out = None
        for _ in range(3):
            del out
            tm = {}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = D.iknn_build(dui, diu, 1.0e-6, None, timing=tm)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            ks.append(tm.get("build_kernel_ms", 0.0))
        chk = int((out.indices.long() * 31 + out.values.view(torch.int32).long()).sum().item()) \
            ^ int(out.indptr.sum().item())
        print(json.dumps({"lib": Path(path).name, "W": int(w), "seconds": round(min(ts), 4),
                          "build+mirror_ms": round(min(ks), 3), "nnz": int(out.indices.shape[0]),
                          "checksum": chk}), flush=True)
        del out, dui, diu
        torch.cuda.empty_cache()
