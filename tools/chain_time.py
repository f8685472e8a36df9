#!/usr/bin/env python3
"""Standalone time of the reference-order chain kernel (csrc/als_rhs.hip) on the ML-25M shape: one
hybrid item-half plan, LK_ALS_SIDE_STREAM=0 (serial launches), HIP-event time of half-epochs with
the chains minus the same plan in accurate mode is not separable -- so this tool just runs N
half-epochs under rocprofv3 --kernel-trace and leaves the per-kernel durations to the trace:

    LK_ALS_SIDE_STREAM=0 rocprofv3 --kernel-trace -d out -o t -- python tools/chain_time.py
"""
import os
import sys
from pathlib import Path

import numpy as np
import scipy.sparse as sps

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    import torch

    from lkpy_amd import synth
    from lkpy_amd._als_engine import HipBackend, ImplicitALSEngine

    k = int(os.environ.get("K", "64"))
    dev = torch.device("cuda:0")
    r = synth.ml25m_like()
    ui = sps.csr_array((np.full(r.nnz, 40.0, np.float32), r.indices, r.indptr), shape=r.shape)
    eng = ImplicitALSEngine(ui, k, 0.1, 0.1, None, None, HipBackend(k, dev))
    for _ in range(6):
        eng.train_epoch()
    eng.check()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
