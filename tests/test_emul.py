"""
The lane-level NumPy models of the ALS Cholesky kernels (tools/emul) stay runnable: they are
how the index arithmetic of csrc/als_chol.hip (accumulator-tile layout, L image, permlane
transposition) is checked without a GPU.  Likewise the LDS placement of the DMA-staged top-K
filter kernel (csrc/topk.hip::score_filter64_kernel).
"""
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools" / "emul"))


@pytest.mark.parametrize("model", ["hybrid_chol", "panel_chol"])
@pytest.mark.parametrize("kp", [16, 64])
def test_lane_level_model_solves(model, kp):
    mod = __import__(model)
    rng = np.random.default_rng(kp)
    m = rng.standard_normal((kp + 30, kp)).astype(np.float32)
    a = (m.T @ m + 0.5 * np.eye(kp)).astype(np.float32)
    y = rng.standard_normal(kp).astype(np.float32)
    x, minpiv = mod.solve(a, y)
    ref = np.linalg.solve(a.astype(np.float64), y.astype(np.float64))
    assert minpiv > 0
    assert np.linalg.norm(x - ref) <= 1e-5 * np.linalg.norm(ref)


def test_permlane_transposition_model():
    import hybrid_chol as h

    x = [np.arange(64, dtype=np.float32) + 100 * r for r in range(4)]
    y = h.transpose4(x)
    lanes = np.arange(64)
    for r in range(4):
        # y[r] at row group g = x[g] at row group r
        want = 100 * (lanes >> 4) + (16 * r + (lanes & 15))
        assert np.array_equal(y[r], want.astype(np.float32))


def test_filter64_lds_layout_model():
    "every operand fetch of score_filter64_kernel reads what the LDS DMA placed; 2-way banks"
    import filter64_layout as f

    assert f.check() == (2, 2)
