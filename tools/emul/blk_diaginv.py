#!/usr/bin/env python3
"""
NUMERICS prototype for the next step of the k = 128 / 256 row solve: a
blocked right-looking Cholesky whose 16 x 16 diagonal blocks are factored once and INVERTED, so
that the panel rows, the forward step and the back substitution become small GEMMs / mat-vecs
(MFMA work without the 16-step `v_readlane` chain that is 27 k of the 164 k cycles of a k = 128
row today, and without the redundant factorisation of the diagonal block in every wave).

Explicit inverses of the diagonal blocks are less stable than substitution; this script puts a
number on it BEFORE any kernel is written: everything below runs in float32 with the
accumulation order of the kernels (products summed in k order), on the rows of the
reference-generated fixtures (tests/golden/als_ref_rows.npz: well- and ill-conditioned normal
matrices, n = 1 ... 40 000 entries), and reports the error of

    sub      the blocked Cholesky with substitution (what csrc/als_blk.hip does today)
    diaginv  the same with inverted diagonal blocks

against the float64 solution and against the reference's own float32 output.

    python tools/emul/blk_diaginv.py [k ...]
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

f32 = np.float32
NB = 16


def _chol_unblocked(a: np.ndarray) -> np.ndarray:
    "lower Cholesky of a small SPD block, float32, column by column"
    n = a.shape[0]
    L = np.zeros_like(a)
    a = a.copy()
    for j in range(n):
        d = f32(np.sqrt(a[j, j]))
        L[j, j] = d
        L[j + 1:, j] = (a[j + 1:, j] / d).astype(f32)
        a[j + 1:, j + 1:] = (a[j + 1:, j + 1:] - np.outer(L[j + 1:, j], L[j + 1:, j])).astype(f32)
    return L


def _tri_inv(L: np.ndarray) -> np.ndarray:
    "inverse of a lower-triangular block by forward substitution on the identity, float32"
    n = L.shape[0]
    X = np.zeros_like(L)
    for c in range(n):
        e = np.zeros(n, f32)
        e[c] = 1
        for i in range(c, n):
            s = e[i]
            for j in range(c, i):
                s = f32(s - f32(L[i, j] * X[j, c]))
            X[i, c] = f32(s / L[i, i])
    return X


def _mm(a, b):
    return (a.astype(f32) @ b.astype(f32)).astype(f32)


def solve_blocked(A: np.ndarray, y: np.ndarray, diaginv: bool):
    "A x = y by blocked Cholesky (block NB) in float32; diaginv: inverted diagonal blocks"
    k = A.shape[0]
    A = A.astype(f32).copy()
    b = y.astype(f32).copy()
    L = np.zeros_like(A)
    inv = {}
    for j0 in range(0, k, NB):
        j1 = min(j0 + NB, k)
        L11 = _chol_unblocked(A[j0:j1, j0:j1])
        L[j0:j1, j0:j1] = L11
        if j1 < k:
            if diaginv:
                Li = _tri_inv(L11)
                inv[j0] = Li
                L21 = _mm(A[j1:, j0:j1], Li.T)  # panel rows: one small GEMM
            else:
                # substitution: solve L21 L11^T = A21 column by column
                L21 = np.zeros((k - j1, j1 - j0), f32)
                A21 = A[j1:, j0:j1]
                for c in range(j1 - j0):
                    s = A21[:, c].copy()
                    for t in range(c):
                        s = (s - L21[:, t] * L11[c, t]).astype(f32)
                    L21[:, c] = (s / L11[c, c]).astype(f32)
            L[j1:, j0:j1] = L21
            A[j1:, j1:] = (A[j1:, j1:] - _mm(L21, L21.T)).astype(f32)
        elif diaginv:
            inv[j0] = _tri_inv(L11)
    # forward: L z = b
    z = np.zeros(k, f32)
    for j0 in range(0, k, NB):
        j1 = min(j0 + NB, k)
        if diaginv:
            z[j0:j1] = _mm(inv[j0], b[j0:j1, None])[:, 0]
        else:
            for i in range(j0, j1):
                s = b[i]
                for t in range(j0, i):
                    s = f32(s - f32(L[i, t] * z[t]))
                z[i] = f32(s / L[i, i])
        if j1 < k:
            b[j1:] = (b[j1:] - _mm(L[j1:, j0:j1], z[j0:j1, None])[:, 0]).astype(f32)
    # backward: L^T x = z
    x = np.zeros(k, f32)
    for j0 in reversed(range(0, k, NB)):
        j1 = min(j0 + NB, k)
        r = z[j0:j1].copy()
        if j1 < k:
            r = (r - _mm(L[j1:, j0:j1].T, x[j1:, None])[:, 0]).astype(f32)
        if diaginv:
            x[j0:j1] = _mm(inv[j0].T, r[:, None])[:, 0]
        else:
            for i in reversed(range(j0, j1)):
                s = r[i - j0]
                for t in range(i + 1, j1):
                    s = f32(s - f32(L[t, i] * x[t]))
                x[i] = f32(s / L[i, i])
    return x


def fixture_rows(ks, kinds=("centered", "skewed"), ns=(5, 64, 100, 1000, 5000)):
    root = Path(__file__).resolve().parent.parent.parent / "tests" / "golden"
    sys.path.insert(0, str(root))
    import als_fixture_inputs as fx

    ref = np.load(root / "als_ref_rows.npz")
    for kind in kinds:
        for k in ks:
            for n in ns:
                case = fx.RowCase(kind, k, n)
                emb = fx.embeddings(case)
                items, vals = fx.row_entries(case)
                otor = ref[f"otor_{kind}_k{k}"]
                M = emb[items].astype(np.float64)
                A = otor.astype(np.float64) + (M.T * vals.astype(np.float64)) @ M
                y = M.T @ (vals.astype(np.float64) + 1.0)
                yield case, A, y, ref[f"x_{case.name}"]


def main(ks):
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))  # noqa: E731
    print(f"{'case':26s} {'cond':>9s} {'sub vs f64':>11s} {'inv vs f64':>11s} {'ref vs f64':>11s}"
          f" {'sub vs ref':>11s} {'inv vs ref':>11s}")
    worst = 0.0
    for case, A, y, xref in fixture_rows(ks):
        x64 = np.linalg.solve(A, y)
        xs = solve_blocked(A, y, False)
        xi = solve_blocked(A, y, True)
        cond = np.linalg.cond(A)
        print(f"{case.name:26s} {cond:9.2e} {rel(xs, x64):11.2e} {rel(xi, x64):11.2e} "
              f"{rel(xref, x64):11.2e} {rel(xs, xref):11.2e} {rel(xi, xref):11.2e}")
        worst = max(worst, rel(xi, x64) / max(rel(xs, x64), 1e-9))
    print(f"worst ratio diaginv / substitution error vs float64: {worst:.2f}")
    return worst


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [128])
