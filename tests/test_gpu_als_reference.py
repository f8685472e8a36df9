"""
GPU parity against the REFERENCE's own outputs (not via the oracle): the HIP half-epoch
kernels run on the inputs of ``tests/golden/als_ref_*.npz`` and are compared with what the
reference's ``_train_new_row`` / ``solve_cholesky`` produced for them
(``tests/golden/make_als_fixtures.py``; ``src/lenskit/als/_implicit.py:97-130``).

Tolerance: the north-star 1e-4 relative wherever the conditioning allows a float32 solve to meet
it -- ``cond(A) * 2^-24 * sqrt(n) < 2e-4`` -- and the forward-error bound
``0.5 * cond * u * sqrt(n) + 1e-6`` otherwise (the same bound the CPU oracle is held to in
``tests/test_oracle_pinned.py``).  Every k the library serves: 25 and 64 (``als_chol.hip``),
128 and 256 (``als_blk.hip``; rows of <= 64 entries through the Woodbury kernels when the plan
enables them).
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu

GOLD = Path(__file__).resolve().parent / "golden"
sys.path.insert(0, str(GOLD))
import als_fixture_inputs as fx  # noqa: E402

U32 = 2.0**-24


def _rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


@pytest.mark.parametrize("kind", ["centered", "skewed"])
@pytest.mark.parametrize("k,wb", [(25, False), (64, False), (128, False), (128, True),
                                  (256, False), (256, True)])
def test_rows_against_reference(gpu, monkeypatch, kind, k, wb):
    """all 13 row lengths of one (kind, k) in ONE half-epoch launch: 13 rows + an empty one;
    ``wb``: rows of <= 64 entries through the Woodbury kernels (forced on for this tiny matrix)"""
    from lkpy_amd import _device as D
    from lkpy_amd import _native

    monkeypatch.setenv("LK_ALS_WB_MIN_ROWS", "1" if wb else "0")

    rows = np.load(GOLD / "als_ref_rows.npz")
    cases = [c for c in fx.row_cases() if c.kind == kind and c.k == k]
    emb = fx.embeddings(cases[0])
    ents = [fx.row_entries(c) for c in cases]
    lens = [len(i) for i, _ in ents] + [0]
    indptr = np.zeros(len(lens) + 1, np.int64)
    np.cumsum(lens, out=indptr[1:])
    mat = sps.csr_array((np.concatenate([v for _, v in ents]), np.concatenate([i for i, _ in ents]),
                         indptr), shape=(len(lens), fx.N_CATALOGUE))
    otor = rows[f"otor_{kind}_k{k}"]

    csr = D.DeviceCSR.from_arrays(mat.indptr.astype(np.int32), mat.indices, mat.data, mat.shape, gpu)
    plan = D.ALSPlan(csr, k, _native.SOLVER_CHOLESKY)
    assert plan.use_wb == wb
    d_this = D.to_device_padded(np.ones((len(lens), k), np.float32), gpu)
    d_other = D.to_device_padded(emb, gpu)
    import torch

    d_otor = torch.from_numpy(np.ascontiguousarray(otor)).to(gpu)  # the reference's own OtOr
    assert _rel(D.Gramian(k, gpu)(d_other, fx.REG).cpu().numpy(), otor) < 1e-5
    plan.half_epoch(d_this, d_other, d_otor)
    plan.check_status()
    got = D.to_host_unpadded(d_this, k)
    assert not got[-1].any()  # the empty row: zeros (implicit.rs:98-101)

    within = 0
    for r, c in enumerate(cases):
        want = rows[f"x_{c.name}"]
        items, vals = ents[r]
        M = emb[items].astype(np.float64)
        A = otor.astype(np.float64) + (M.T * vals.astype(np.float64)) @ M
        cond = float(np.linalg.cond(A))
        e = _rel(got[r], want)
        scale = cond * U32 * np.sqrt(c.n)
        if scale < 2e-4:
            assert e <= 1.0e-4, (c.name, e, cond)
        assert e <= 0.5 * scale + 1e-6, (c.name, e, scale, cond)
        within += e <= 1e-4
    print(f"{kind} k={k} wb={wb}: {within}/{len(cases)} GPU rows within 1e-4 of the reference's own row solve")


@pytest.mark.parametrize("half", ["P1", "Q1", "P3", "Q3"])
def test_ml_small_half_epochs_against_reference(gpu, oracle, half):
    """cfg1 half-epochs from the reference's own states.  ml-latest-small is ill-conditioned
    (cond 1e3 ... 2e5): rows are held to 16 * cond * u + 2e-6 (the CPU oracle meets the same
    vectors with the constant 4) and to 1e-4 where cond * u < 1e-5."""
    from lkpy_amd import _device as D
    from lkpy_amd import _native

    g = np.load(GOLD / "als_ref_mlsmall.npz")
    ui, iu = fx.ml_small_matrices()
    rng = np.random.default_rng(fx.ML_SEED)
    Q0 = oracle.als_initial_params(rng, ui.shape[1], fx.ML_K)
    csr, other = {"P1": (ui, Q0), "Q1": (iu, g["P1"]), "P3": (ui, g["Q2"]), "Q3": (iu, g["P3"])}[half]
    want = g[half]
    dcsr = D.DeviceCSR.from_arrays(csr.indptr.astype(np.int32), csr.indices, csr.data, csr.shape, gpu)
    plan = D.ALSPlan(dcsr, fx.ML_K, _native.SOLVER_CHOLESKY)
    d_this = D.to_device_padded(np.zeros_like(want), gpu)
    d_other = D.to_device_padded(np.ascontiguousarray(other), gpu)
    d_otor = D.Gramian(fx.ML_K, gpu)(d_other, fx.ML_REG)
    plan.half_epoch(d_this, d_other, d_otor)
    plan.check_status()
    got = D.to_host_unpadded(d_this, fx.ML_K)
    _x64, cond = oracle.als_referee_f64(csr, other, fx.ML_REG)
    num = np.linalg.norm(got.astype(np.float64) - want, axis=1)
    den = np.linalg.norm(want.astype(np.float64), axis=1)
    assert not got[den == 0].any()
    e = np.where(den > 0, num / np.maximum(den, 1e-300), 0.0)
    cu = cond * U32
    # (the CPU oracle meets this with the constant 4; cond is a lower-bound estimate and the
    # a-priori constant of a float32 Cholesky solve is O(k): 16 for the GPU's different order)
    assert (e <= 16.0 * cu + 2.0e-6).all(), float((e / np.maximum(cu, 1e-30)).max())
    # the raw north-star tolerance wherever it is decidable between two float32 solves of the same
    # system (cond u < 2.5e-5: VERDICT r4 item 3); the other rows are counted, not hidden
    decid = (cu < 2.5e-5) & (den > 0)
    assert (e[decid] <= 1e-4).all(), float(e[decid].max())
    rest = (~decid) & (den > 0)
    print(f"{half}: GPU vs reference rel {_rel(got, want):.2e}; max err/(cond u) "
          f"{float((e[cond > 0] / cu[cond > 0]).max()):.2f}; rows with cond u < 2.5e-5: "
          f"{int(decid.sum())}, all within 1e-4 (max {float(e[decid].max()) if decid.any() else 0:.1e}); "
          f"other rows {int(rest.sum())}, of those over 1e-4: {int((e[rest] > 1e-4).sum())} "
          f"(max {float(e[rest].max()) if rest.any() else 0:.1e})")


@pytest.mark.parametrize("kind", ["centered", "skewed"])
@pytest.mark.parametrize("k", fx.ROW_K)
def test_explicit_rows_against_reference(gpu, kind, k):
    """The explicit (biased-MF) half-epoch kernels against the reference's own
    ``_train_bias_row_cholesky`` outputs (``als_ref_explicit.npz``; src/lenskit/als/_explicit.py:
    121-147): 12 row lengths + an empty row in one launch; every row within 1e-4."""
    from lkpy_amd import _device as D
    from lkpy_amd import _native

    gold = np.load(GOLD / "als_ref_explicit.npz")
    cases = [c for c in fx.explicit_cases() if c.kind == kind and c.k == k]
    emb = fx.embeddings(cases[0])
    items = [fx.row_entries(c)[0] for c in cases]
    vals = [fx.explicit_values(c) for c in cases]
    lens = [len(i) for i in items] + [0]
    indptr = np.zeros(len(lens) + 1, np.int64)
    np.cumsum(lens, out=indptr[1:])
    mat = sps.csr_array((np.concatenate(vals), np.concatenate(items), indptr),
                        shape=(len(lens), fx.N_CATALOGUE))
    csr = D.DeviceCSR.from_arrays(mat.indptr.astype(np.int32), mat.indices, mat.data, mat.shape, gpu)
    plan = D.ALSPlan(csr, k, _native.SOLVER_CHOLESKY)
    d_this = D.to_device_padded(np.ones((len(lens), k), np.float32), gpu)
    d_other = D.to_device_padded(emb, gpu)
    plan.half_epoch_explicit(d_this, d_other, fx.REG)
    plan.check_status()
    got = D.to_host_unpadded(d_this, k)
    assert not got[-1].any()  # explicit.rs:91-94: empty row -> zeros
    worst = 0.0
    for r, c in enumerate(cases):
        e = _rel(got[r], gold[f"x_{c.name}"])
        worst = max(worst, e)
        assert e <= 1.0e-4, (c.name, e)
    print(f"explicit {kind} k={k}: {len(cases)}/{len(cases)} GPU rows within 1e-4 of the reference's "
          f"own row solve (worst {worst:.1e})")
