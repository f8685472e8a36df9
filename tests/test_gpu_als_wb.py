"""
Woodbury row solve for short rows at large k (csrc/als_wb.hip): the same half-epoch through
`lk_als_implicit_half_epoch`, rows with <= 16 (16 x 16 system, csrc/als_wb.hip) and with
17 .. 64 entries (64 x 64 system, als_wb64_kernel in csrc/als_chol.hip) taking the rank-n path, against the
oracle's dense `sposv` restatement (src/accel/als/implicit.rs:87-125) and against this
library's own dense kernel.  Tolerance: 1e-4 relative (north star).
"""
import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu

RTOL = 1e-4


def _short_csr(rng, n_rows, n_cols, varied_values):
    u = rng.random(n_rows)
    lens = np.where(u < 0.6, rng.integers(0, 17, n_rows),
                    np.where(u < 0.9, rng.integers(17, 65, n_rows), rng.integers(65, 300, n_rows)))
    lens[:65] = np.arange(65)  # every length 0..64 present
    lens[65:129] = np.arange(65, 129)  # ... and 65..128 (als_wb128_kernel at padded k = 256)
    indptr = np.zeros(n_rows + 1, np.int64)
    np.cumsum(lens, out=indptr[1:])
    indices = np.empty(indptr[-1], np.int32)
    for r in range(n_rows):
        indices[indptr[r]:indptr[r + 1]] = np.sort(
            rng.choice(n_cols, lens[r], replace=False)).astype(np.int32)
    values = np.full(indptr[-1], 40.0, np.float32)
    if varied_values:  # use_ratings: weight * rating, incl. zero confidence increments
        values = (rng.integers(0, 11, indptr[-1]) * 4.0).astype(np.float32)
    return sps.csr_array((values, indices, indptr), shape=(n_rows, n_cols))


@pytest.mark.parametrize("is64", [False, True])
@pytest.mark.parametrize("k,varied", [(100, False), (128, True), (200, False), (256, True)])
def test_short_rows_vs_oracle_and_dense(gpu, oracle, rng, monkeypatch, k, varied, is64):
    from lkpy_amd import _device as D
    from lkpy_amd import _native

    n_rows, n_cols = 3000, 4000
    mat = _short_csr(rng, n_rows, n_cols, varied)
    other = (rng.standard_normal((n_cols, k)) * 0.1).astype(np.float32)
    this = (rng.standard_normal((n_rows, k)) * 0.1).astype(np.float32)
    otor = oracle.implicit_otor(other, 0.1)
    want = this.copy()
    want_frob = oracle.als_half_epoch(mat, want, other, otor)

    indptr = mat.indptr.astype(np.int64 if is64 else np.int32)
    csr = D.DeviceCSR.from_arrays(indptr, mat.indices, mat.data, mat.shape, gpu)
    d_other = D.to_device_padded(other, gpu)
    d_otor = D.Gramian(k, gpu)(d_other, 0.1)

    def run(min_rows):
        monkeypatch.setenv("LK_ALS_WB_MIN_ROWS", str(min_rows))
        plan = D.ALSPlan(csr, k, _native.SOLVER_CHOLESKY)
        d_this = D.to_device_padded(this, gpu)
        frob = plan.half_epoch(d_this, d_other, d_otor)
        plan.check_status()
        return plan, D.to_host_unpadded(d_this, k), float(frob.item()), d_this

    plan, got, frob, d_this = run(1)
    assert plan.use_wb and plan.short_rows >= 1500
    # the 64 x 64 and 128 x 128 variants only pay at padded k = 256
    wb_max = 128 if plan.kp == 256 else 64  # (k = 128: LK_ALS_WB64_K128 = 64 by default)
    plan0, dense, frob0, _ = run(0)
    assert not plan0.use_wb

    lens = np.diff(mat.indptr)
    assert np.all(got[lens == 0] == 0.0)  # implicit.rs:98-101
    # rows the Woodbury kernels did not touch are bit-identical to the dense run
    assert np.array_equal(got[lens > wb_max], dense[lens > wb_max])
    # ... with the 128 x 128 variant switched off, so are the rows with 65 .. 128 entries
    monkeypatch.setenv("LK_ALS_WB128", "0")
    _, got64, _, _ = run(1)
    monkeypatch.delenv("LK_ALS_WB128")
    assert np.array_equal(got64[lens > 64], dense[lens > 64])
    assert np.array_equal(got64[lens <= 64], got[lens <= 64])
    # ... and with the 64 x 64 variant switched off, so are the rows with 17 .. 64 entries
    monkeypatch.setenv("LK_ALS_WB64", "0")
    _, got16, _, _ = run(1)
    monkeypatch.delenv("LK_ALS_WB64")
    assert np.array_equal(got16[lens > 16], dense[lens > 16])
    assert np.array_equal(got16[lens <= 16], got[lens <= 16])
    for name, ref in (("oracle", want), ("dense kernel", dense)):
        rn = np.linalg.norm(ref, axis=1)
        err = np.linalg.norm(got - ref, axis=1)
        assert np.all(err <= 5 * RTOL * np.maximum(rn, 1e-3)), name
        assert np.linalg.norm(got - ref) <= RTOL * np.linalg.norm(ref), name
    assert abs(frob - want_frob) <= 1e-4 * want_frob
    # closer to (or as close as) the float64 solution as the reference arithmetic
    exact = oracle.als_half_epoch_f64(mat, other, 0.1)
    short = (lens > 0) & (lens <= wb_max)
    e_gpu = np.linalg.norm((got - exact)[short]) / np.linalg.norm(exact[short])
    e_ref = np.linalg.norm((want - exact)[short]) / np.linalg.norm(exact[short])
    assert e_gpu <= max(2 * e_ref, 2e-6), (e_gpu, e_ref)
    # pad columns stay zero, bit-reproducible
    if d_this.shape[1] > k:
        assert float(d_this[:, k:].abs().max().item()) == 0.0
    _, again, _, _ = run(1)
    assert np.array_equal(again, got)


@pytest.mark.parametrize("k", [100, 128, 200, 256])
def test_device_spd_inverse_matches_float64(gpu, rng, k):
    """``lk_spd_inverse`` (csrc/spd_inverse.hip: float32 register sweep + two float64
    Newton-Schulz steps) against NumPy's float64 inverse, rounded to float32 -- what the rounds 1-2
    ``torch.linalg.cholesky_ex`` + ``cholesky_inverse`` call produced on the host side."""
    import ctypes

    import torch

    from lkpy_amd import _device as D
    from lkpy_amd import _native

    lib = _native.require_gpu()
    kp = D.padded_dim(k)
    for cond_kind in ("init", "trained"):
        if cond_kind == "init":  # the first epoch's factors: G ~ reg I (cond ~ 1)
            m = ((rng.standard_normal((50_000, k)) * 0.01) ** 2).astype(np.float32)
        else:  # positive-mean factors: one dominant eigenvalue, cond ~ 1e4
            m = (rng.random((50_000, k)) * 0.3).astype(np.float32)
        g = (m.astype(np.float64).T @ m.astype(np.float64) + 0.1 * np.eye(k)).astype(np.float32)
        want = np.linalg.inv(g.astype(np.float64))
        d_g = torch.from_numpy(g).to(gpu)
        out = torch.full((kp, kp), 7.0, dtype=torch.float32, device=gpu)
        flag = torch.full((1,), -1, dtype=torch.int32, device=gpu)
        ws = torch.empty(lib.lk_spd_inverse_workspace_bytes(k), dtype=torch.uint8, device=gpu)
        _native.check(lib.lk_spd_inverse(D._ptr(d_g), k, k, D._ptr(out), D._ptr(flag), D._ptr(ws),
                                         D._stream()), "lk_spd_inverse")
        got = out.cpu().numpy()
        assert int(flag.item()) == 0
        assert not got[k:, :].any() and not got[:, k:].any()  # zero padding
        err = np.abs(got[:k, :k] - want).max() / np.abs(want).max()
        cond = np.linalg.cond(g.astype(np.float64))
        print(f"k={k} {cond_kind}: cond {cond:.1e}, max rel err of the float32 result {err:.1e}")
        assert err < 2e-7  # one float32 rounding of the float64 inverse
    # not positive definite: flag set, zeros out
    bad = np.eye(k, dtype=np.float32)
    bad[k // 2, k // 2] = -1.0
    _native.check(lib.lk_spd_inverse(D._ptr(torch.from_numpy(bad).to(gpu)), k, k, D._ptr(out),
                                     D._ptr(flag), D._ptr(ws), D._stream()), "lk_spd_inverse")
    assert int(flag.item()) != 0 and not out.cpu().numpy().any()


@pytest.mark.parametrize("k", [128, 256])
def test_otor_not_positive_definite_takes_the_dense_fallback_on_the_device(gpu, oracle, rng,
                                                                           monkeypatch, k):
    """reg = 0 with rank-deficient factors: OtOr has no inverse, so Z does not exist -- the
    Woodbury kernels stand down and the dense fallback launch solves their rows, decided on the
    device (``status[1]``).  Rows whose own matrix IS positive definite come out as the dense
    kernel computes them; here every non-empty row's matrix is singular too (rank <= n + rank(G)
    < k), so the half-epoch reports the solve error ``sposv`` would (implicit.rs:79)."""
    from lkpy_amd import _device as D
    from lkpy_amd import _native

    n_rows, n_cols = 600, 40  # 40 factor rows: rank(G) <= 40 < k
    lens = rng.integers(0, 31, n_rows)
    lens[:31] = np.arange(31)
    indptr = np.zeros(n_rows + 1, np.int64)
    np.cumsum(lens, out=indptr[1:])
    indices = np.concatenate([np.sort(rng.choice(n_cols, ln, replace=False)) for ln in lens])
    mat = sps.csr_array((np.full(indptr[-1], 40.0, np.float32), indices.astype(np.int32), indptr),
                        shape=(n_rows, n_cols))
    other = (rng.standard_normal((n_cols, k)) * 0.1).astype(np.float32)
    this = (rng.standard_normal((n_rows, k)) * 0.1).astype(np.float32)
    csr = D.DeviceCSR.from_arrays(mat.indptr.astype(np.int32), mat.indices, mat.data, mat.shape, gpu)
    d_other = D.to_device_padded(other, gpu)
    d_otor = D.Gramian(k, gpu)(d_other, 0.0)  # reg = 0: singular

    def run(min_rows):
        monkeypatch.setenv("LK_ALS_WB_MIN_ROWS", str(min_rows))
        plan = D.ALSPlan(csr, k, _native.SOLVER_CHOLESKY)
        d_this = D.to_device_padded(this, gpu)
        plan.half_epoch(d_this, d_other, d_otor)
        try:
            plan.check_status()
            err = None
        except RuntimeError as e:
            err = str(e)
        return plan, D.to_host_unpadded(d_this, k), err

    plan, got, err = run(1)
    assert plan.use_wb
    plan0, dense, err0 = run(0)
    assert not plan0.use_wb
    # the same outcome as the all-dense run: same error (or none), same rows bit for bit
    assert (err is None) == (err0 is None)
    if err is not None:
        assert "ALS solve error" in err and "ALS solve error" in err0
    assert np.all(got[lens == 0] == 0.0) and np.all(dense[lens == 0] == 0.0)
    # every non-empty row's matrix is singular in both runs: whatever sposv-like garbage the
    # solves leave is reported through the status word, not compared


def test_row_slices_with_negative_values_keep_the_dense_kernels(gpu, oracle, rng, monkeypatch):
    """three row slices, each with fewer short rows than LK_ALS_WB_MIN_ROWS
    (so no slice looks at the values on its own) but more in sum (so the GROUP wants the Woodbury
    kernels), over a matrix with negative confidence values (use_ratings with negative ratings):
    the group must find them and keep the dense kernels -- sqrt(v) of a negative value would be
    NaN -- and the result must be the oracle's dense sposv solution."""
    from lkpy_amd import _device as D
    from lkpy_amd import _native

    k, n_rows, n_cols = 128, 1800, 2500
    mat = _short_csr(rng, n_rows, n_cols, True)
    mat.data[rng.random(mat.nnz) < 0.1] *= -0.05  # a few small negative increments (A stays SPD)
    assert mat.data.min() < 0
    other = (rng.standard_normal((n_cols, k)) * 0.1).astype(np.float32)
    this = (rng.standard_normal((n_rows, k)) * 0.1).astype(np.float32)
    want = this.copy()
    oracle.als_half_epoch(mat, want, other, oracle.implicit_otor(other, 0.1))

    full = D.DeviceCSR.from_arrays(mat.indptr.astype(np.int32), mat.indices, mat.data, mat.shape,
                                   gpu)
    lens = np.diff(mat.indptr)
    cuts = [(0, 600), (600, 1200), (1200, 1800)]
    # (padded k = 128: the Woodbury kernels take rows of up to 64 entries, LK_ALS_WB64_K128)
    short = [int(((lens[lo:hi] <= 64)).sum()) for lo, hi in cuts]
    monkeypatch.setenv("LK_ALS_WB_MIN_ROWS", str(max(short) + 1))
    assert sum(short) >= max(short) + 1
    plans = []
    for lo, hi in cuts:
        view = D.DeviceCSR(full.indptr[lo:hi + 1], full.indices, full.values, (hi - lo, n_cols),
                           full.h_indptr[lo:hi + 1])
        plans.append(D.ALSPlan(view, k, _native.SOLVER_CHOLESKY))
        assert not plans[-1].use_wb and plans[-1]._negative_values is None  # nobody looked yet
    group = D.ALSPlanGroup(plans, n_cols)
    assert not group.use_wb and all(p.negative_values and not p.use_wb for p in plans)

    d_other = D.to_device_padded(other, gpu)
    d_otor = D.Gramian(k, gpu)(d_other, 0.1)
    d_this = D.to_device_padded(this, gpu)
    for (lo, hi), p in zip(cuts, plans):
        p.half_epoch(d_this[lo:hi], d_other, d_otor)
    group.check_status()
    got = D.to_host_unpadded(d_this, k)
    assert np.isfinite(got).all()
    rn = np.linalg.norm(want, axis=1)
    assert np.all(np.linalg.norm(got - want, axis=1) <= 5 * RTOL * np.maximum(rn, 1e-3))

    # the same slices over non-negative values: the group does switch the Woodbury kernels on
    mat.data[:] = np.abs(mat.data)
    full2 = D.DeviceCSR.from_arrays(mat.indptr.astype(np.int32), mat.indices, mat.data, mat.shape,
                                    gpu)
    plans2 = [D.ALSPlan(D.DeviceCSR(full2.indptr[lo:hi + 1], full2.indices, full2.values,
                                    (hi - lo, n_cols), full2.h_indptr[lo:hi + 1]), k,
                        _native.SOLVER_CHOLESKY) for lo, hi in cuts]
    assert D.ALSPlanGroup(plans2, n_cols).use_wb


@pytest.mark.parametrize("k", [128, 256])
def test_side_stream_inverse_changes_nothing(gpu, rng, monkeypatch, k):
    """OtOr^-1 on the plan's side stream (under the chunk kernel, csrc/als_blk.hip) against the
    same work on the launch stream (`LK_ALS_SIDE_STREAM=0`): bit-identical half-epochs, also when
    repeated (the fork / join events are reused)."""
    from lkpy_amd import _device as D
    from lkpy_amd import _native

    n_cols = 20000
    lens = np.concatenate([rng.integers(0, 17, 3000), rng.integers(17, 200, 500), [2500, 5000]])
    indptr = np.zeros(len(lens) + 1, np.int64)
    np.cumsum(lens, out=indptr[1:])
    indices = np.concatenate(
        [np.sort(rng.choice(n_cols, n, replace=False)) for n in lens]).astype(np.int32)
    mat = sps.csr_array((np.full(indptr[-1], 40.0, np.float32), indices, indptr),
                        shape=(len(lens), n_cols))
    other = (rng.standard_normal((n_cols, k)) * 0.1).astype(np.float32)
    csr = D.DeviceCSR.from_arrays(mat.indptr.astype(np.int32), mat.indices, mat.data, mat.shape, gpu)
    d_other = D.to_device_padded(other, gpu)
    d_otor = D.Gramian(k, gpu)(d_other, 0.1)
    monkeypatch.setenv("LK_ALS_WB_MIN_ROWS", "1")
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("LK_ALS_SIDE_STREAM", mode)
        plan = D.ALSPlan(csr, k, _native.SOLVER_CHOLESKY)
        assert plan.use_wb
        d_this = D.to_device_padded(np.zeros((mat.shape[0], k), np.float32), gpu)
        for _ in range(3):
            plan.half_epoch(d_this, d_other, d_otor)
        plan.check_status()
        got[mode] = D.to_host_unpadded(d_this, k)
    assert np.isfinite(got["1"]).all()
    assert np.array_equal(got["1"], got["0"])


@pytest.mark.parametrize("limit", ["0", "32", "64"])
def test_k128_woodbury_range_knob(gpu, oracle, rng, monkeypatch, limit):
    """Padded k = 128: `LK_ALS_WB64_K128` = 0 / 32 / 64 sends rows of up to 16 / 32 / 64 entries
    through the Woodbury kernels; every setting within 1e-4 of the oracle's dense sposv, and the
    rows outside the range bit-identical to the all-dense run."""
    from lkpy_amd import _device as D
    from lkpy_amd import _native

    k, n_cols = 128, 6000
    lens = np.concatenate([rng.integers(0, 100, 2500), rng.integers(100, 400, 100)])
    indptr = np.zeros(len(lens) + 1, np.int64)
    np.cumsum(lens, out=indptr[1:])
    indices = np.concatenate(
        [np.sort(rng.choice(n_cols, n, replace=False)) for n in lens]).astype(np.int32)
    mat = sps.csr_array((np.full(indptr[-1], 40.0, np.float32), indices, indptr),
                        shape=(len(lens), n_cols))
    other = (rng.standard_normal((n_cols, k)) * 0.1).astype(np.float32)
    this = np.zeros((len(lens), k), np.float32)
    want = this.copy()
    oracle.als_half_epoch(mat, want, other, oracle.implicit_otor(other, 0.1))
    csr = D.DeviceCSR.from_arrays(mat.indptr.astype(np.int32), mat.indices, mat.data, mat.shape, gpu)
    d_other = D.to_device_padded(other, gpu)
    d_otor = D.Gramian(k, gpu)(d_other, 0.1)

    def run(min_rows):
        monkeypatch.setenv("LK_ALS_WB_MIN_ROWS", str(min_rows))
        plan = D.ALSPlan(csr, k, _native.SOLVER_CHOLESKY)
        d_this = D.to_device_padded(this, gpu)
        plan.half_epoch(d_this, d_other, d_otor)
        plan.check_status()
        return plan, D.to_host_unpadded(d_this, k)

    monkeypatch.setenv("LK_ALS_WB64_K128", limit)
    plan, got = run(1)
    _, dense = run(0)
    top = max(16, int(limit))
    assert plan.use_wb and plan.woodbury_rows == int((lens <= top).sum())
    assert np.array_equal(got[lens > top], dense[lens > top])
    assert not np.array_equal(got[(lens > 0) & (lens <= top)], dense[(lens > 0) & (lens <= top)])
    err = np.linalg.norm(got - want, axis=1)
    assert np.all(err <= 5 * RTOL * np.maximum(np.linalg.norm(want, axis=1), 1e-3))
