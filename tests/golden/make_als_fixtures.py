#!/usr/bin/env python3
"""Pin the ALS oracle to the REFERENCE ITSELF, executed here.

The reference package cannot be imported under this image's Python 3.10 (PEP 695 syntax in
other modules, missing wheels: SURVEY.md section 8c) -- but the four functions that hold the ALS
row algebra are plain NumPy / SciPy and parse fine.  This script takes their *source from the
read-only checkout at run time* (``ast`` extraction: nothing is copied into this repository),
executes them, and commits only the resulting input / output VECTORS:

* ``ImplicitMFScorer._train_new_row``   ``src/lenskit/als/_implicit.py:101-130``
* ``solve_cholesky``                    ``src/lenskit/math/solve.py:17-41``
* ``_implicit_otor``                    ``src/lenskit/als/_implicit.py:177-184``
* ``ImplicitMFTrainer.initial_params``  ``src/lenskit/als/_implicit.py:152-155``
* ``_train_bias_row_cholesky``          ``src/lenskit/als/_explicit.py:121-147`` (explicit model)
* ``BiasedMFTrainer.initial_params``    ``src/lenskit/als/_explicit.py:104-108``

Run ONCE in the build container (``/root/reference`` does not exist on the GPU box), with the
BLAS pool pinned to one thread (done below: the summation order of a threaded sgemv depends on the
pool size and the host's load):

    python tests/golden/make_als_fixtures.py

Outputs under ``tests/golden/``:

* ``als_ref_rows.npz``  -- synthetic single-row solves, n = 1 ... 40 000 entries, k = 25 / 64 /
  128 / 256, constant and varied confidence values: the reference's ``x`` (and its ``OtOr``) for
  inputs that ``als_fixture_inputs.py`` regenerates from integer hashes (no RNG-stream
  dependence);
* ``als_ref_explicit.npz`` -- the explicit (biased-MF) row solve ``_train_bias_row_cholesky`` on
  the same synthetic rows with bias-normalised ratings in [-2.5, 2.5), reg 0.1 (A = M^T M +
  reg n I, rhs M^T r), and the first rows of ``BiasedMFTrainer.initial_params`` (unit rows);
* ``als_ref_mlsmall.npz`` -- ml-latest-small (cfg1: k = 25, weight 40, reg 0.1, seed 42):
  the reference's ``initial_params`` draws (items first), then three epochs in which every row
  is solved BY THE REFERENCE'S OWN ``_train_new_row`` / ``solve_cholesky`` with ``OtOr`` from its
  own ``_implicit_otor`` (the epoch order and the zero rows for empty items follow
  ``src/accel/als/implicit.rs:56-125`` and ``src/lenskit/als/_common.py:241-256``).  Stored:
  P1, Q1 (first epoch from the initial state), Q2 (input of the third epoch), P3, Q3 and the
  Frobenius deltas -- so each half-epoch of the oracle can be checked FROM IDENTICAL INPUTS.

``tests/test_oracle_pinned.py`` checks ``oracle/lk_oracle.c`` (the restatement of the Rust
kernel the GPU is compared against) and ``oracle/lk_oracle.py`` against these vectors.
"""
from __future__ import annotations

import __future__ as _future
import ast
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import scipy.linalg
import scipy.sparse as sps

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(OUT))
import als_fixture_inputs as fx  # noqa: E402


def _extract(path: Path, name: str, cls: str | None = None):
    "compile ONE function definition of the reference file, annotations left unevaluated"
    tree = ast.parse(path.read_text())
    body = tree.body
    if cls is not None:
        body = next(n for n in body if isinstance(n, ast.ClassDef) and n.name == cls).body
    fn = next(n for n in body if isinstance(n, ast.FunctionDef) and n.name == name)
    fn.decorator_list = []  # @override
    mod = ast.Module(body=[fn], type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = {"np": np, "cho_factor": scipy.linalg.cho_factor, "cho_solve": scipy.linalg.cho_solve}
    code = compile(mod, f"<{path.relative_to(REF)}:{fn.lineno}>", "exec",
                   flags=_future.annotations.compiler_flag, dont_inherit=True)
    exec(code, ns)
    return ns[name], ns, (fn.lineno, fn.end_lineno)


def reference_functions():
    imp = REF / "src/lenskit/als/_implicit.py"
    solve, _, l_s = _extract(REF / "src/lenskit/math/solve.py", "solve_cholesky")
    row, ns_row, l_r = _extract(imp, "_train_new_row", "ImplicitMFScorer")
    ns_row["solve_cholesky"] = solve
    otor, _, l_o = _extract(imp, "_implicit_otor")
    init, _, l_i = _extract(imp, "initial_params", "ImplicitMFTrainer")
    me = SimpleNamespace(logger=SimpleNamespace(debug=lambda *a, **k: None))
    exp = REF / "src/lenskit/als/_explicit.py"
    erow, ns_erow, l_e = _extract(exp, "_train_bias_row_cholesky")
    ns_erow["solve_cholesky"] = solve
    einit, _, l_ei = _extract(exp, "initial_params", "BiasedMFTrainer")
    lines = {"solve_cholesky": l_s, "_train_new_row": l_r, "_implicit_otor": l_o,
             "initial_params": l_i, "_train_bias_row_cholesky": l_e, "explicit initial_params": l_ei}
    fns = SimpleNamespace(explicit_row=erow,
                          explicit_init=lambda rng, n, k: einit(SimpleNamespace(rng=rng), n, k))
    return (lambda items, vals, emb, OtOr: row(me, items, vals, emb, OtOr)), otor, \
        (lambda rng, n, k: init(SimpleNamespace(rng=rng), n, k)), lines, fns


def half_epoch(row_fn, csr: sps.csr_array, this: np.ndarray, other: np.ndarray, otor: np.ndarray):
    "one half-epoch with the reference's row function; returns the f64 Frobenius delta"
    sq = 0.0
    for r in range(csr.shape[0]):
        lo, hi = csr.indptr[r], csr.indptr[r + 1]
        if lo == hi:  # implicit.rs:98-101: empty row -> zeros, no delta
            this[r] = 0.0
            continue
        x = row_fn(csr.indices[lo:hi], csr.data[lo:hi], other, otor)
        assert x.dtype == np.float32
        d = x.astype(np.float64) - this[r]
        sq += float(d @ d)
        this[r] = x
    return np.sqrt(sq)


def main():
    row_fn, otor_fn, init_fn, lines, xfns = reference_functions()
    print("reference functions extracted at lines", lines)

    # ---- synthetic single rows ---------------------------------------------------------
    out = {}
    for case in fx.row_cases():
        emb = fx.embeddings(case)
        items, vals = fx.row_entries(case)
        OtOr = otor_fn(emb, np.float32(case.reg))
        x = row_fn(items, vals, emb, OtOr)
        assert x.dtype == np.float32 and OtOr.dtype == np.float32
        out[f"x_{case.name}"] = x
        # OtOr is the same for every n of one (k, kind): keep one copy
        out.setdefault(f"otor_{case.kind}_k{case.k}", OtOr)
    np.savez_compressed(OUT / "als_ref_rows.npz", **out)
    print("als_ref_rows.npz:", len(out), "arrays")

    # ---- explicit model: _train_bias_row_cholesky ---------------------------------------
    out = {}
    for case in fx.explicit_cases():
        emb = fx.embeddings(case)
        items, _ = fx.row_entries(case)
        x = xfns.explicit_row(items, fx.explicit_values(case), emb, np.float32(case.reg))
        assert x.dtype == np.float32
        out[f"x_{case.name}"] = x
    init = xfns.explicit_init(np.random.default_rng(fx.ML_SEED), 64, 25)
    out["init_head"] = init
    np.savez_compressed(OUT / "als_ref_explicit.npz", **out)
    print("als_ref_explicit.npz:", len(out), "arrays")

    # ---- EASE: the reference's own SPD inverse ``_chol_invert_torch`` (knn/ease.py:190-209) ----
    import threading

    import torch

    inv_fn, ns_inv, l_inv = _extract(REF / "src/lenskit/knn/ease.py", "_chol_invert_torch")
    ns_inv.update(torch=torch, _chol_lock=threading.Lock())
    cooc = fx.ease_cooc(OUT / "ml_small.npz")
    cooc[np.diag_indices(len(cooc))] += np.float32(fx.EASE_REG)  # ease.py:113-114
    inv = inv_fn(cooc.copy(), device="cpu")
    assert inv.dtype == np.float32
    np.savez_compressed(OUT / "ease_ref_inverse.npz", rows=fx.EASE_ROWS,
                        inverse_rows=inv[fx.EASE_ROWS], diag=np.diag(inv).copy())
    print("ease_ref_inverse.npz: _chol_invert_torch at lines", l_inv, "on a",
          cooc.shape, "co-occurrence matrix")

    # ---- ml-latest-small, cfg1 ---------------------------------------------------------
    ui, iu = fx.ml_small_matrices(OUT / "ml_small.npz")
    U, I = ui.shape
    rng = np.random.default_rng(fx.ML_SEED)
    Q = init_fn(rng, I, fx.ML_K)  # items first (_common.py:291-294)
    P = init_fn(rng, U, fx.ML_K)
    res = {"Q0_head": Q[:8].copy(), "P0_head": P[:8].copy(),
           "Q0_sum": np.float64(Q.astype(np.float64).sum()), "P0_sum": np.float64(P.astype(np.float64).sum())}
    deltas = []
    for ep in range(1, 4):
        if ep == 3:
            res["Q2"] = Q.copy()
        du = half_epoch(row_fn, ui, P, Q, otor_fn(Q, np.float32(fx.ML_REG)))
        di = half_epoch(row_fn, iu, Q, P, otor_fn(P, np.float32(fx.ML_REG)))
        deltas.append((du, di))
        if ep in (1, 3):
            res[f"P{ep}"] = P.copy()
            res[f"Q{ep}"] = Q.copy()
        print(f"epoch {ep}: deltaP {du:.5f} deltaQ {di:.5f}")
    res["deltas"] = np.asarray(deltas)
    res["OtOr3"] = otor_fn(Q, np.float32(fx.ML_REG))  # _save_user_otor (_implicit.py:171-175)
    np.savez_compressed(OUT / "als_ref_mlsmall.npz", **res)
    print("als_ref_mlsmall.npz written")


if __name__ == "__main__":
    # ONE BLAS thread while the reference's functions run: OpenBLAS splits a long sgemv / sgemm
    # reduction over its pool, and how it splits depends on the thread count and on the load of
    # the host -- with the pool pinned the vectors are a function of the inputs alone, and
    # tests/test_oracle_pinned.py (which pins the pool the same way) can hold the NumPy half of
    # the oracle to them BIT FOR BIT on any host (the unpinned check failed
    # once in three runs on a busy host).
    from threadpoolctl import threadpool_limits

    with threadpool_limits(limits=1, user_api="blas"):
        main()
