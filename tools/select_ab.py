"""A/B of the selection tier boundary of the fused top-N on a 10 000-user batch recommend call:
LK_TOPK_LONG_EXCL (exclusion entries beyond which a row goes to the second tier) and
LK_TOPK_SPLIT.  python tools/select_ab.py"""
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from lkpy_amd import _device as D  # noqa: E402
from lkpy_amd import synth  # noqa: E402

r = synth.ml25m_like()
nu, ni = r.shape
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
Q = torch.randn(ni, 64, device=dev, generator=g) * 0.1
csr = D.DeviceCSR.from_arrays(r.indptr, r.indices, np.zeros(0, np.float32), r.shape, dev)
csr.values = None
users = np.random.default_rng(11).choice(nu, 10000, replace=False).astype(np.int32)
hist = D.gather_rows(csr, users, scale=40.0)
U = torch.randn(10000, 64, device=dev, generator=g) * 0.1
print("longest history in the batch:", int(np.diff(hist.h_indptr).max()))
ref = None
for split in ("1", "0"):
    for le in ("4096", "1024", "256", "100000"):
        os.environ["LK_TOPK_SPLIT"], os.environ["LK_TOPK_LONG_EXCL"] = split, le
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            idx, sc = D.score_topk(U, Q, 64, 100, hist.indptr, hist.indices)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        chk = (int(idx.sum().item()), float(sc.sum().item()))
        ref = ref or chk
        print(f"split {split} long_excl {le:>6}: {best * 1e3:.3f} ms  same lists: {chk == ref}")
