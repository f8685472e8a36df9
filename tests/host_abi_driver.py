#!/usr/bin/env python3
"""
Runs INTEGRATION.md's binding sketch AS WRITTEN, in a fresh interpreter with raw ``ctypes``: no
``lkpy_amd`` import, no torch.  Driven by ``tests/test_gpu_host_abi.py`` (needs a GPU); prints one
JSON object.  The oracle (test infrastructure) is the checker.

``lk_als_implicit_half_epoch_host[_ctl]`` replaces ``train_implicit_matrix``
(src/accel/als/implicit.rs:35-84) with its task controls (src/accel/tasks/mod.rs:62-106).
"""
from __future__ import annotations

import json
import re
import sys
import threading
import time
import types
from pathlib import Path

import numpy as np
import pyarrow as pa
import scipy.sparse as sps

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def load_sketch():
    text = (ROOT / "INTEGRATION.md").read_text()
    m = re.search(r"```python\n(# lenskit/_accel_amd\.py.*?)```", text, flags=re.S)
    assert m, "INTEGRATION.md lost its binding sketch"
    mod = types.ModuleType("lenskit_accel_amd_sketch")
    exec(compile(m.group(1), "INTEGRATION.md:sketch", "exec"), mod.__dict__)
    return mod


def matrix_of(csr: sps.csr_array, is64: bool):
    "what the sketch reads: .offsets / .indices / .values with .to_numpy() (SparseRowArray's)"
    off = csr.indptr.astype(np.int64 if is64 else np.int32)
    return types.SimpleNamespace(offsets=pa.array(off), indices=pa.array(csr.indices.astype(np.int32)),
                                 values=pa.array(csr.data.astype(np.float32)))


def make_case(rng, k, n_rows, n_cols, long_len, short_rows=0):
    lens = rng.integers(1, 300, n_rows)
    lens[0] = long_len            # a hybrid-order row (> 2048 entries): chunks + chain
    lens[1] = 0                   # an empty row (implicit.rs:98-101)
    lens[2] = 1
    lens[3] = 2048
    if short_rows:
        lens[-short_rows:] = rng.integers(1, 17, short_rows)
    indptr = np.zeros(n_rows + 1, np.int64)
    np.cumsum(lens, out=indptr[1:])
    idx = np.concatenate([np.sort(rng.choice(n_cols, ln, replace=False)) for ln in lens])
    mat = sps.csr_array((np.full(indptr[-1], 40.0, np.float32), idx.astype(np.int32), indptr),
                        shape=(n_rows, n_cols))
    other = (np.abs(rng.standard_normal((n_cols, k))) * 0.05).astype(np.float32)
    other[rng.random((n_cols, k)) < 0.3] *= -1.0
    this = (rng.standard_normal((n_rows, k)) * 0.01).astype(np.float32)
    return mat, this, other


def row_rel(got, want):
    num = np.linalg.norm(got.astype(np.float64) - want, axis=1)
    den = np.linalg.norm(want.astype(np.float64), axis=1)
    return np.where(den > 0, num / np.maximum(den, 1e-300), num)


def main():
    from oracle import lk_oracle as lko

    sk = load_sketch()
    assert "lkpy_amd" not in sys.modules and "torch" not in sys.modules
    out = {"cases": []}
    rng = np.random.default_rng(7)
    for k, is64 in ((25, False), (64, True), (128, False), (256, True)):
        mat, this, other = make_case(rng, k, 96 if k > 64 else 160, 6000, 5000)
        otor = lko.implicit_otor(other, 0.1)
        want = this.copy()
        want_frob = lko.als_half_epoch(mat, want, other, otor)
        got = this.copy()
        task = sk.train_implicit_matrix(matrix_of(mat, is64), got, other, otor)
        res = {}
        th = threading.Thread(target=lambda: res.update(frob=task.invoke()))  # off the main thread
        th.start()
        th.join()
        rel = row_rel(got, want)
        out["cases"].append({
            "k": k, "offsets": 64 if is64 else 32, "rows": int(mat.shape[0]),
            "row_rel_max": float(rel.max()), "long_row_rel": float(rel[0]),
            "empty_row_zero": bool(not got[1].any()), "frob": res["frob"], "want_frob": float(want_frob),
            "progress": list(task.current_progress()),
        })

    # the Woodbury path of the plain host entry (no task control): 6000 short rows at k = 128
    k = 128
    mat, this, other = make_case(rng, k, 6200, 6000, 3000, short_rows=6000)
    otor = lko.implicit_otor(other, 0.1)
    want = this.copy()
    lko.als_half_epoch(mat, want, other, otor)
    got = this.copy()
    frob = sk.ctypes.c_float()
    lib = sk._lib
    lib.lk_als_implicit_half_epoch_host.restype = sk.ctypes.c_int
    lib.lk_als_implicit_half_epoch_host.argtypes = lib.lk_als_implicit_half_epoch_host_ctl.argtypes[:-1]
    off = mat.indptr.astype(np.int32)
    idx32, val32 = mat.indices.astype(np.int32), mat.data.astype(np.float32)
    rc = lib.lk_als_implicit_half_epoch_host(
        off.ctypes.data, 0, idx32.ctypes.data, val32.ctypes.data, mat.shape[0],
        mat.shape[1], k, got.ctypes.data, other.ctypes.data, otor.ctypes.data, 2,
        sk.ctypes.byref(frob))
    out["woodbury_host"] = {"rc": int(rc), "row_rel_max": float(row_rel(got, want).max()),
                            "error": lib.lk_last_error().decode() if rc else ""}

    # a non-positive-definite otor: RuntimeError("ALS solve error: ...") (implicit.rs:79)
    mat, this, other = make_case(rng, 64, 64, 6000, 2100)
    bad = -np.eye(64, dtype=np.float32) * 1.0e4
    try:
        sk.train_implicit_matrix(matrix_of(mat, False), this.copy(), other, bad).invoke()
        out["not_spd"] = "no error"
    except RuntimeError as e:
        out["not_spd"] = str(e)
    # wrong array type: TypeError before any device work
    try:
        sk.train_implicit_matrix(matrix_of(mat, False), this.astype(np.float64), other, bad)
        out["type_error"] = False
    except TypeError:
        out["type_error"] = True

    # progress + cancel through the task object: a half-epoch long enough to be observed
    k = 128
    n_rows, n_cols = 120_000, 20_000
    lens = rng.integers(60, 140, n_rows)
    indptr = np.zeros(n_rows + 1, np.int64)
    np.cumsum(lens, out=indptr[1:])
    idx = rng.integers(0, n_cols, indptr[-1]).astype(np.int32)
    big = sps.csr_array((np.full(indptr[-1], 40.0, np.float32), idx, indptr), shape=(n_rows, n_cols))
    other = (np.abs(rng.standard_normal((n_cols, k))) * 0.05).astype(np.float32)
    otor = lko.implicit_otor(other, 0.1)
    this = np.zeros((n_rows, k), np.float32)
    task = sk.train_implicit_matrix(matrix_of(big, True), this, other, otor)
    seen, res = [], {}
    th = threading.Thread(target=lambda: res.update(frob=task.invoke()))
    th.start()
    while th.is_alive():
        seen.append(task.current_progress()[0])
        time.sleep(0.0005)
    th.join()
    out["progress"] = {"final": list(task.current_progress()), "polls": len(seen),
                       "mid_flight_values": int(sum(0 < s < n_rows for s in seen)),
                       "monotone": bool(all(b >= a for a, b in zip(seen, seen[1:]))),
                       "finite": bool(np.isfinite(this).all()), "frob": res.get("frob")}
    this2 = np.zeros((n_rows, k), np.float32)
    task2 = sk.train_implicit_matrix(matrix_of(big, True), this2, other, otor)
    err = []

    def work():
        try:
            task2.invoke()
        except BaseException as e:  # noqa: BLE001
            err.append(type(e).__name__)

    th2 = threading.Thread(target=work)
    th2.start()
    while th2.is_alive() and task2.current_progress()[0] == 0:
        time.sleep(0.0002)
    task2.cancel()
    th2.join()
    done = task2.current_progress()[0]
    solved = int((np.abs(this2).sum(axis=1) > 0).sum())
    out["cancel"] = {"error": err, "rows_done": int(done), "rows_total": n_rows,
                     "rows_written": solved}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
