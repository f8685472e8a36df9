#!/usr/bin/env python3
"""
Lane-level NumPy model of the hybrid Cholesky of als_chol.hip (LK_ALS_PANEL=2): panels of four
columns are factored in the lane = row layout (registers + v_readlane only), turned into MFMA
operands by a 4 x 4 (register x row-group) transposition made of v_permlane32_swap /
v_permlane16_swap, and the trailing update is one v_mfma_f32_16x16x4_f32 per tile.
    python tools/emul/hybrid_chol.py
"""
import numpy as np

from panel_chol import LANES, SLOT, SUB, c0, mfma_16x16x4, off, tidx


def permlane32_swap(a, b):
    "lanes 32-63 of a <-> lanes 0-31 of b"
    a, b = a.copy(), b.copy()
    t = a[32:].copy()
    a[32:] = b[:32]
    b[:32] = t
    return a, b


def permlane16_swap(a, b):
    "odd rows (16 lanes) of a <-> even rows of b"
    a, b = a.copy(), b.copy()
    for r in (0, 2):
        t = a[16 * (r + 1):16 * (r + 2)].copy()
        a[16 * (r + 1):16 * (r + 2)] = b[16 * r:16 * (r + 1)]
        b[16 * r:16 * (r + 1)] = t
    return a, b


def transpose4(x):
    "y[r] at row group g = x[g] at row group r"
    x0, x2 = permlane32_swap(x[0], x[2])
    x1, x3 = permlane32_swap(x[1], x[3])
    y0, y1 = permlane16_swap(x0, x1)
    y2, y3 = permlane16_swap(x2, x3)
    return [y0, y1, y2, y3]


def solve(A, y):
    KP = A.shape[0]
    NT = KP // 16
    f = np.float32
    T = {}
    for tj in range(NT):
        for ti in range(tj + 1):
            t = np.zeros((64, 4), f)
            for r in range(4):
                t[:, r] = A[16 * ti + 4 * SLOT + r, 16 * tj + SUB]
            T[tidx(ti, tj)] = t
    b = np.where(LANES < KP, y[np.minimum(LANES, KP - 1)], 0).astype(f)  # lane = row
    SIZE = KP * KP // 2 + KP
    img = np.full(SIZE, np.nan, f)
    rinvarr = np.zeros(KP, f)
    scr = np.zeros(4 * 64, f)
    minpiv = f(3e38)
    for m in range(KP // 4):
        J, tj, mg = 4 * m, m >> 2, m & 3
        # extraction: lane (t, c) <- registers 0..3 of lane (mg, c) of tile (tj, t)
        for t in range(tj, NT):
            w = SLOT == mg
            for r in range(4):
                scr[((t * 16 + SUB) * 4 + r)[w]] = T[tidx(tj, t)][w, r]
        pr = [scr[(SLOT * 16 + SUB) * 4 + s].copy() for s in range(4)]
        live = (SLOT >= tj) & (LANES < KP)
        pr = [np.where(live, v, f(0)).astype(f) for v in pr]
        lp = [None] * 4
        for s0 in range(4):
            j = J + s0
            piv = pr[s0][j]
            minpiv = min(minpiv, piv)
            rinv = f(1.0) / np.sqrt(piv, dtype=f)
            rinvarr[j] = rinv
            lj = np.where(LANES > j, pr[s0] * rinv, f(0)).astype(f)
            w = (LANES >= c0(j)) & (LANES < KP)
            img[(off(j, KP) + LANES - c0(j))[w]] = lj[w]
            zj = b[j] * rinv
            b = b - lj * zj
            for s in range(s0 + 1, 4):
                pr[s] = pr[s] - lj * lj[J + s]
            lp[s0] = lj
        q = transpose4(lp)
        for ti in range(tj, NT):
            for t2 in range(ti, NT):
                T[tidx(ti, t2)] = mfma_16x16x4(-q[ti], q[t2], T[tidx(ti, t2)])
    lane = LANES
    dinv = np.where(lane < KP, rinvarr[np.minimum(lane, KP - 1)], 0).astype(f)
    b = b * dinv  # z
    my_c0 = (lane + 1) & ~3
    for j4 in range(KP // 4 - 1, -1, -1):
        l4 = np.zeros((64, 4), f)
        for i in range(64):
            if i < KP - 1 and 4 * j4 >= my_c0[i]:
                base = off(i, KP) - my_c0[i] + 4 * j4
                l4[i] = img[base:base + 4]
        for u in range(3, -1, -1):
            j = 4 * j4 + u
            if j >= 1:
                xj = (b * dinv)[j]
                b = b - l4[:, u] * xj
    b = b * dinv
    assert not np.isnan(b[:KP]).any(), "read of an unwritten L-image cell"
    return b[:KP], minpiv


def main():
    rng = np.random.default_rng(3)
    for KP in (16, 32, 64):
        for trial in range(3):
            M = rng.standard_normal((KP + 40, KP)).astype(np.float32)
            A = (M.T @ M + 0.5 * np.eye(KP)).astype(np.float32)
            y = rng.standard_normal(KP).astype(np.float32)
            x, mp = solve(A, y)
            ref = np.linalg.solve(A.astype(np.float64), y.astype(np.float64))
            err = np.linalg.norm(x - ref) / np.linalg.norm(ref)
            print(KP, trial, "rel err %.2e" % err, "minpiv %.3g" % mp)
            assert err < 1e-4


if __name__ == "__main__":
    main()
