"""``lenskit._accel.als`` stand-in (src/lenskit/_accel/als.pyi:11-16)."""
from __future__ import annotations

import numpy as np
import torch

from .. import _device as D
from .. import _native
from ..parallel import AccelTask
from ._util import as_csr_arrays


def train_implicit_matrix(matrix, this: np.ndarray, other: np.ndarray, otor: np.ndarray,
                          *, solver: int = _native.SOLVER_AUTO) -> AccelTask[float]:
    """
    One implicit-ALS half-epoch (src/accel/als/implicit.rs:35-84): ``this`` ([rows x k] f32,
    C-contiguous, writeable) is UPDATED IN PLACE; returns a task yielding
    sqrt(sum ||delta row||^2).  Solver failure -> RuntimeError("ALS solve error: ...").
    """
    if not (isinstance(this, np.ndarray) and this.dtype == np.float32 and
            this.flags.c_contiguous and this.flags.writeable):
        raise TypeError("this must be a writeable C-contiguous float32 array")
    offsets, indices, values, shape = as_csr_arrays(matrix)
    if values is None:
        raise TypeError("train_implicit_matrix needs a sparse matrix with values")
    rows, k = this.shape
    other = np.ascontiguousarray(other, dtype=np.float32)
    otor = np.ascontiguousarray(otor, dtype=np.float32)
    assert shape == (rows, other.shape[0]) and other.shape[1] == k and otor.shape == (k, k)

    def run(task: AccelTask) -> float:
        dev = D.device()
        csr = D.DeviceCSR.from_arrays(offsets, indices, values, shape, dev)
        plan = D.ALSPlan(csr, k, solver)
        ctl = D.TaskCtl()
        plan.set_ctl(ctl)
        task.attach(ctl)  # cancel() / current_progress() now reach the running kernels
        d_this = D.to_device_padded(this, dev)
        d_other = D.to_device_padded(other, dev)
        d_otor = torch.from_numpy(otor).to(dev)
        frob = plan.half_epoch(d_this, d_other, d_otor)
        plan.check_status()  # KeyboardInterrupt if cancelled, RuntimeError if a solve failed
        this[...] = D.to_host_unpadded(d_this, k)
        task.set_progress(rows)
        return float(frob.item())

    return AccelTask(run, total=rows)
