#!/usr/bin/env python3
"""
Lane-level NumPy emulation (one wave = 64 lanes, lane = 16*slot + sub) of the FIRST panel
Cholesky tried for als_chol.hip: everything in the accumulator-tile layout, finished columns
broadcast with ds_bpermute.  Correct, but measured slower than the lane = row solver (the
bpermute round trip sits in every column's dependent chain) and removed from the kernel; the
shipped scheme is tools/emul/hybrid_chol.py, which shares the helpers defined here
(MFMA / L-image models).
    python tools/emul/panel_chol.py
"""
import numpy as np

LANES = np.arange(64)
SLOT, SUB = LANES >> 4, LANES & 15


def tidx(ti, tj):
    return tj * (tj + 1) // 2 + ti


def mfma_16x16x4(a_op, b_op, d):
    "D[i][j] += sum_k A[i][k] B[k][j]; A from lane (k, i), B from lane (k, j); D lane (s, c) reg r = D[4s+r][c]"
    A = a_op.reshape(4, 16).T  # A[i][k]
    B = b_op.reshape(4, 16)    # B[k][j]
    P = (A.astype(np.float32) @ B.astype(np.float32)).astype(np.float32)
    out = d.copy()
    for r in range(4):
        out[:, r] += P[4 * SLOT + r, SUB]
    return out


def c0(j):
    return (j + 1) & ~3


def off(j, KP):
    m, r = j >> 2, j & 3
    return 4 * KP * m - 8 * m * m + 4 * m + r * (KP - 4 * m)


def solve(A, y):
    "A: primed SPD [KP, KP], y [KP] -> x via the emulated kernel algorithm"
    KP = A.shape[0]
    NT = KP // 16
    f = np.float32
    # accumulator tiles as the Gram phase leaves them
    T = {}
    for tj in range(NT):
        for ti in range(tj + 1):
            t = np.zeros((64, 4), f)
            for r in range(4):
                t[:, r] = A[16 * ti + 4 * SLOT + r, 16 * tj + SUB]
            T[tidx(ti, tj)] = t
    yv = [y[16 * t + SUB].astype(f) for t in range(NT)]  # replicated over slots
    SIZE = KP * KP // 2 + KP
    img = np.full(SIZE, np.nan, f)
    rinvarr = np.zeros(KP, f)
    zarr = np.zeros(KP, f)
    scr = np.zeros(NT * 64, f)
    minpiv = f(3e38)
    for m in range(KP // 4):
        J, tj, mg = 4 * m, m >> 2, m & 3
        jc = 4 * mg
        p = [None] * NT
        for t in range(tj, NT):
            w = SLOT == mg
            for r in range(4):
                scr[((t * 16 + SUB) * 4 + r)[w]] = T[tidx(tj, t)][w, r]
            p[t] = scr[(t * 16 + SUB) * 4 + SLOT].copy()
        for s0 in range(4):
            piv = p[tj][16 * s0 + jc + s0]
            minpiv = min(minpiv, piv)
            rinv = f(1.0) / np.sqrt(piv, dtype=f)
            rinvarr[J + s0] = rinv
            mine = SLOT == s0
            rsel = np.where(mine, rinv, f(1.0)).astype(f)
            # no zeroing above the diagonal: those cells only ever reach dead rows / columns
            for t in range(tj, NT):
                p[t] = p[t] * rsel
            bc = [None] * NT
            for t in range(tj, NT):
                bc[t] = p[t][16 * s0 + SUB]
            if s0 < 3:
                sc = p[tj][np.minimum(16 * s0 + jc + SLOT, 63)]
                scm = np.where(SLOT > s0, sc, f(0)).astype(f)
                for t in range(tj, NT):
                    p[t] = p[t] - bc[t] * scm
            zj = yv[tj][jc + s0] * rinv
            zarr[J + s0] = zj
            for t in range(tj, NT):
                yv[t] = yv[t] - bc[t] * zj
            # strictly-lower L image, column J+s0: rows >= c0 (zeros on and above the diagonal)
            j = J + s0
            for t in range(tj, NT):
                row = 16 * t + SUB
                v = bc[t] if t > tj else np.where(SUB > jc + s0, bc[t], f(0))
                w = (SLOT == 0) & (row >= c0(j))
                img[(off(j, KP) + row - c0(j))[w]] = v[w]
        for ti in range(tj, NT):
            for t2 in range(ti, NT):
                T[tidx(ti, t2)] = mfma_16x16x4(-p[ti], p[t2], T[tidx(ti, t2)])
    # back substitution, lane = primed row (the kernel's existing code)
    lane = LANES
    dinv = np.where(lane < KP, rinvarr[np.minimum(lane, KP - 1)], 0).astype(f)
    b = np.where(lane < KP, zarr[np.minimum(lane, KP - 1)], 0).astype(f)
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
