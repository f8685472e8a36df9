"""
The lane-level NumPy models of the ALS Cholesky kernels (tools/emul) stay runnable: they are
how the index arithmetic of csrc/als_chol.hip (accumulator-tile layout, L image, permlane
transposition) is checked without a GPU.  Likewise the LDS placement of the DMA-staged top-K
filter kernel (csrc/topk.hip::score_filter64_kernel), the per-target accumulator of the kNN
scoring kernels (csrc/iknn_score.hip: rounds, rank sort, BinaryHeap replay, in-place heapify)
against the C oracle, the folded 8-way reduction of the CG kernel (csrc/als_cg.hip), and the
wave-per-row selection of the fused top-N path (csrc/topk.hip::cand_select_wave_kernel).
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


@pytest.mark.parametrize("max_nbrs", [1, 2, 5, 20])
@pytest.mark.parametrize("explicit", [True, False])
def test_knn_accumulator_model_matches_the_oracle(oracle, max_nbrs, explicit):
    """csrc/iknn_score.hip per target, modelled in tools/emul/knn_accum.py: rounds of history rows
    with the hits arriving in arbitrary order, rank sort, vector -> heap (staged pushes as in the
    list kernel, reverse + sift-up in place as in the slot kernel), std's BinaryHeap push / pop,
    sequential unfused sums -- against the C restatement of accum.rs, bit for bit, on inputs
    FULL of equal weights (where the eviction order matters)."""
    import scipy.sparse as sps

    import knn_accum as ka

    rng = np.random.default_rng(100 * max_nbrs + explicit)
    for trial in range(40):
        n_hist = int(rng.integers(0, 90))
        hit = rng.random(n_hist) < 0.7
        # few distinct weights: ties everywhere, at the boundary too
        weights = rng.choice(np.array([0.125, 0.25, 0.25, 0.5, 0.3, 0.7], np.float32), n_hist)
        values = rng.standard_normal(n_hist).astype(np.float32)
        target = n_hist  # one more item: the target
        rows = np.flatnonzero(hit)
        sims = sps.csr_array((weights[rows], (rows, np.full(len(rows), target))),
                             shape=(n_hist + 1, n_hist + 1), dtype=np.float32)
        want_s, want_c = oracle.iknn_score(sims, np.arange(n_hist, dtype=np.int32),
                                           values if explicit else None,
                                           np.array([target], np.int32), max_nbrs, 1)
        hits = [(int(r), weights[r], values[r] if explicit else np.float32(0)) for r in rows]
        for in_place in (False, True):
            for cap in (256, 16, 7):
                got_s, got_c = ka.score_target(hits, max_nbrs, 1, explicit, cap=cap,
                                               in_place=in_place, rng=rng)
                assert got_c == int(want_c[0]), (trial, in_place, cap)
                if np.isnan(want_s[0]):
                    assert np.isnan(got_s)
                else:
                    assert np.float32(got_s).view(np.uint32) == want_s[:1].view(np.uint32)[0], \
                        (trial, in_place, cap, got_s, want_s[0])


def test_cg_reduce8_model():
    "csrc/als_cg.hip::cg_reduce8: lane L ends with the wave-wide sum of item CG_REV3(L & 7)"
    import knn_accum as ka

    rng = np.random.default_rng(8)
    a = rng.standard_normal((64, 8))
    d = ka.cg_reduce8(a)
    tot = a.sum(axis=0)
    for lane in range(64):
        assert abs(d[lane] - tot[ka.cg_rev3(lane & 7)]) < 1e-12
    assert sorted(ka.cg_rev3(j) for j in range(8)) == list(range(8))
    assert all(ka.cg_rev3(ka.cg_rev3(j)) == j for j in range(8))


def test_inverted_diagonal_blocks_are_as_accurate_as_substitution():
    """tools/emul/blk_diaginv.py -- the numerics of the next k = 128 / 256 row solve:
    a blocked float32 Cholesky with INVERTED 16 x 16 diagonal blocks (panel rows and both
    triangular solves as small GEMMs) against the substitution form the kernel uses today."""
    import blk_diaginv as b

    rng = np.random.default_rng(3)
    rel = lambda a, c: float(np.linalg.norm(a - c) / np.linalg.norm(c))  # noqa: E731
    for k, cond in ((48, 1e2), (128, 1e3), (128, 1e5)):
        q, _ = np.linalg.qr(rng.standard_normal((k, k)))
        a = (q * np.geomspace(1.0, cond, k)) @ q.T
        y = rng.standard_normal(k)
        x64 = np.linalg.solve(a, y)
        es, ei = rel(b.solve_blocked(a, y, False), x64), rel(b.solve_blocked(a, y, True), x64)
        u = 2.0 ** -24
        assert es < 2 * cond * u and ei < 2 * cond * u, (k, cond, es, ei)
        assert ei < 3 * es + 1e-7, (k, cond, es, ei)


def test_wave_select_model_equals_a_sort():
    """hash of the candidates' items with struck exclusions, two-stage threshold search, ballot
    compaction, 128-key register network: the top n of what survives, in order"""
    import wave_select as w

    assert w.main(trials=120) > 32  # the index-half search ran (more than 128 equal scores)
