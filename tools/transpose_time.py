"""Device transpose of the ML-25M-shaped matrix (both orientations) and the full ranking of 64
score rows, timed: python tools/transpose_time.py -- the two users of csrc/radix_sort.h."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from lkpy_amd import _device as D  # noqa: E402
from lkpy_amd import synth  # noqa: E402

r = synth.ml25m_like()
dev = torch.device("cuda:0")
csr = D.DeviceCSR.from_arrays(r.indptr, r.indices, r.data, r.shape, dev)
for name, m in (("users x items -> items x users", csr),):
    t = D.csr_transpose(m)
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t = D.csr_transpose(m)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"{name}: {best * 1e3:.2f} ms ({m.nnz} entries)")
    tt = D.csr_transpose(t)
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tt = D.csr_transpose(t)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"and back: {best * 1e3:.2f} ms; identity: "
          f"{bool(torch.equal(tt.indices, m.indices) and torch.equal(tt.indptr, m.indptr))}")
g = torch.Generator(device=dev).manual_seed(1)
sc = torch.randn(64, 62423, device=dev, generator=g)
D.argtopn(sc, -1)
torch.cuda.synchronize()
t0 = time.perf_counter()
out = D.argtopn(sc, -1)
torch.cuda.synchronize()
print(f"full ranking of 64 x 62423 scores: {(time.perf_counter() - t0) * 1e3:.2f} ms; sorted: "
      f"{bool((torch.gather(sc, 1, out.long()).diff(dim=1) <= 0).all())}")
