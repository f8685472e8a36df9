#!/usr/bin/env python3
"""Device -> pageable host transfer of a similarity-matrix-sized result (4.6 GB of int32 column
numbers < 62 423): lk_download variants side by side.  python tools/download_bench.py"""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lkpy_amd import _device as D  # noqa: E402

dev = torch.device("cuda:0")
n = 1_154_165_798
idx = torch.randint(0, 62423, (n,), dtype=torch.int32, device=dev)
ref = None
for thp in ("0", "1"):
    for narrow in ("0", "1"):
        for threads in (8, 16, 30):
            os.environ["LK_DOWNLOAD_THP"] = thp
            os.environ["LK_DOWNLOAD_NARROW"] = narrow
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            h = D.to_host(idx, threads=threads, index_bound=62423)
            dt = time.perf_counter() - t0
            if ref is None:
                ref = idx.cpu().numpy()
            ok = bool(np.array_equal(h, ref))
            del h
            print(json.dumps({"thp": thp, "narrow_u16": narrow, "threads": threads,
                              "seconds": round(dt, 4), "GB_per_s_of_result": round(n * 4 / dt / 1e9, 1),
                              "equal": ok}), flush=True)
