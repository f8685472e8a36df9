"""
The function seam is DROP-IN: the reference's own consumer lines, copied from the call sites
they were written for, run against the ``lkpy_amd._accel`` stand-ins:

* ``ItemKNNScorer.train``  src/lenskit/knn/item.py:161-197 (``compute_similarities``)
* ``ItemKNNScorer.__call__``  src/lenskit/knn/item.py:247-295 (``score_explicit/implicit``)
* ``ItemList.top_n``  src/lenskit/data/_items.py:975-998 (``argtopn`` / ``argsort_descending``)
* ``AccelTask`` protocol  src/lenskit/parallel/_task.py:34-57, src/accel/tasks/mod.rs:62-106
"""
import threading
import time

import numpy as np
import pyarrow as pa
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu


def test_compute_similarities_consumer_lines(gpu, oracle, ml_small):
    from lkpy_amd import _accel
    from lkpy_amd.matrix import SparseRowArray
    from lkpy_amd.parallel import run_accel_task

    ui, iu, _means, _ = oracle.iknn_prepare(ml_small["rmat"], True)
    n_rows, n_items = ui.shape
    # item.py:155-156
    ui_mat = SparseRowArray.from_scipy(ui)
    iu_mat = SparseRowArray.from_scipy(iu)

    class PB:  # item_progress(...) stand-in: run_accel_task calls update(completed=)
        seen = []

        def update(self, completed=None, **kw):
            self.seen.append(completed)

    pb = PB()
    # item.py:161-171
    smat = run_accel_task(
        _accel.knn.compute_similarities(ui_mat, iu_mat, (n_rows, n_items), 1.0e-6, None),
        progress=pb,
    )
    # item.py:173-177
    assert isinstance(smat, list)
    smat = pa.chunked_array(smat)
    smat = smat.combine_chunks()
    assert pa.types.is_large_list(smat.type)
    smat = SparseRowArray.from_array(smat)
    # item.py:182-197
    lengths = np.diff(smat.offsets)
    assert np.sum(lengths > 0) > 0
    assert smat.offsets[-1].as_py() == len(smat.values), f"{smat.offsets[-1]} != {len(smat.values)}"
    item_counts = np.diff(smat.offsets.to_numpy())

    want = oracle.iknn_build(ui, iu, 1.0e-6, None)
    assert smat.shape == (n_items, n_items)
    assert np.array_equal(item_counts, np.diff(want.indptr))
    assert np.array_equal(smat.indices.to_numpy(), want.indices)
    assert np.array_equal(smat.values.to_numpy().view(np.uint32), want.data.view(np.uint32))

    # raw (non-extension) Arrow input straight from the C Data Interface side is accepted too
    raw_ui, raw_iu = ui_mat.storage, iu_mat.storage
    assert isinstance(raw_ui, pa.ListArray)
    again = run_accel_task(
        _accel.knn.compute_similarities(raw_ui, raw_iu, (n_rows, n_items), 1.0e-6, 20))
    trunc = SparseRowArray.from_array(pa.chunked_array(again).combine_chunks())
    want20 = oracle.iknn_build(ui, iu, 1.0e-6, 20)
    assert np.array_equal(trunc.offsets.to_numpy(), want20.indptr)
    assert np.array_equal(trunc.values.to_numpy().view(np.uint32), want20.data.view(np.uint32))
    with pytest.raises(TypeError):
        _accel.knn.compute_similarities(pa.array([1, 2, 3]), iu_mat, (n_rows, n_items), 1e-6, None)


@pytest.mark.parametrize("explicit", [True, False])
def test_score_consumer_lines(gpu, oracle, ml_small, explicit, rng):
    from lkpy_amd import _accel
    from lkpy_amd.matrix import SparseRowArray

    rmat = ml_small["rmat"]
    if not explicit:
        rmat = sps.coo_array((np.ones(rmat.nnz, np.float32), (rmat.row, rmat.col)), rmat.shape)
    ui, iu, means, _ = oracle.iknn_prepare(rmat, explicit)
    sims = oracle.iknn_build(ui, iu, 1.0e-6, None)
    sim_matrix = SparseRowArray.from_scipy(sims, large=True)
    item_means = None if means is None else np.asarray(means).ravel()
    csr = sps.csr_array(rmat)
    u = 17
    # a history with one item the model does not know (-1) and targets with unknowns
    ri_nums = np.concatenate([csr.indices[csr.indptr[u]:csr.indptr[u + 1]], [-1]]).astype(np.int32)
    ratings = np.concatenate([csr.data[csr.indptr[u]:csr.indptr[u + 1]], [3.0]]).astype(np.float32)
    ti_nums = np.concatenate([rng.choice(ui.shape[1], 300, replace=False), [-1, -1]]).astype(np.int32)
    max_nbrs, min_nbrs = 20, 2

    # ---- src/lenskit/knn/item.py:247-291, verbatim up to names ----
    ri_mask = ri_nums >= 0
    ri_arr = pa.array(ri_nums, mask=~ri_mask)
    ti_mask = ti_nums >= 0
    ti_arr = pa.array(ti_nums, mask=~ti_mask)
    if explicit:
        ri_vals = ratings.astype(np.float32, copy=True)
        ri_vals[ri_mask] -= item_means[ri_nums[ri_mask]]
        ri_vals = pa.array(ri_vals, mask=~ri_mask)
        scores, counts = _accel.knn.score_explicit(sim_matrix, ri_arr, ri_vals, ti_arr,
                                                   max_nbrs, min_nbrs)
        scores = scores.to_numpy(zero_copy_only=False, writable=True)
        scores[ti_mask] += item_means[ti_nums[ti_mask]]
    else:
        scores, counts = _accel.knn.score_implicit(sim_matrix, ri_arr, ti_arr, max_nbrs, min_nbrs)
        scores = scores.to_numpy(zero_copy_only=False, writable=True)
    # ----
    assert isinstance(counts, pa.Int32Array) and counts.null_count == 2  # the two null targets
    assert scores.dtype == np.float32 and np.isnan(scores[-2:]).all()

    rr = None
    if explicit:
        rr = ratings.copy()
        rr[ri_mask] -= item_means[ri_nums[ri_mask]]
    ws, wc = oracle.iknn_score(sims, ri_nums[ri_mask], None if rr is None else rr[ri_mask],
                               ti_nums, max_nbrs, min_nbrs)
    if explicit:
        ws[ti_mask] += item_means[ti_nums[ti_mask]]
    assert np.array_equal(np.isnan(scores), np.isnan(ws))
    ok = ~np.isnan(ws)
    assert np.allclose(scores[ok], ws[ok], rtol=1e-5, atol=1e-6)
    assert np.array_equal(counts.fill_null(-1).to_numpy(), wc)
    with pytest.raises(TypeError):
        _accel.knn.score_implicit(sim_matrix, pa.array(["a"]), ti_arr, max_nbrs, min_nbrs)


def test_top_n_consumer_lines_beyond_4096(gpu, oracle, rng):
    "ItemList.top_n (data/_items.py:975-998): want_all and n > 4096 on 9 125 scored items"
    from lkpy_amd import _accel as _data_accel_pkg

    _data_accel = _data_accel_pkg.data
    n_items = 9125  # ml-latest-small's item count: `recommender` without n ranks them all
    vals = rng.standard_normal(n_items).astype(np.float32)
    vals[rng.choice(n_items, 500, replace=False)] = np.nan
    vals[100:140] = vals[100]  # ties
    for n in (None, -1, 5000, 9125, 20000, 10):
        scores = pa.array(vals)  # MTArray(scores).arrow()
        want_all = n is None or n < 0
        if want_all:
            picked = _data_accel.argsort_descending(scores)
        else:
            picked = _data_accel.argtopn(scores, min(n, n_items))
        assert isinstance(picked, pa.Int32Array)
        picked = np.asarray(picked)  # self._take(picked, ordered=True)
        valid = int(np.sum(~np.isnan(vals)))
        exp_len = valid if want_all else min(n, valid)
        assert len(picked) == exp_len
        got = vals[picked]
        assert not np.isnan(got).any() and np.all(np.diff(got) <= 0)
        # nothing excluded beats the minimum (tests/accel/test_argsort.py:83-87)
        rest = np.setdiff1d(np.flatnonzero(~np.isnan(vals)), picked)
        if len(rest) and len(picked):
            assert vals[rest].max() <= got[-1]
        # our tie rule (lower index first) makes the list exactly the stable descending sort
        assert np.array_equal(picked, oracle.argsort_descending(vals)[:exp_len])
    # nulls in the Arrow array are not candidates (sorting.rs:143)
    mask = np.zeros(n_items, bool)
    mask[::3] = True
    arr = pa.array(vals, mask=mask)
    v2 = vals.copy()
    v2[mask] = np.nan
    assert np.array_equal(np.asarray(_data_accel.argsort_descending(arr)),
                          oracle.argsort_descending(v2))
    # integer scores (argsort_int / argtopn integer branch)
    ints = pa.array(rng.integers(-1000, 1000, 6000).astype(np.int32))
    p = np.asarray(_data_accel.argtopn(ints, 4500))
    assert len(p) == 4500 and np.all(np.diff(ints.to_numpy()[p]) <= 0)
    with pytest.raises(TypeError):
        _data_accel.argtopn(pa.array(["x"]), 1)


def test_batched_full_sort_matches_selection(gpu, rng):
    "lk_argtopn: the sort path (n > 4096 / n < 0) and the selection kernel agree on prefixes"
    import torch

    from lkpy_amd import _device as D

    s = rng.standard_normal((7, 6000)).astype(np.float32)
    s[:, ::11] = np.nan
    s[2, :] = np.nan  # a row without candidates
    d = torch.from_numpy(s).to(gpu)
    full = D.argtopn(d, -1).cpu().numpy()
    top = D.argtopn(d, 200).cpu().numpy()
    big = D.argtopn(d, 5000).cpu().numpy()
    assert full.shape == (7, 6000) and big.shape == (7, 5000)
    assert np.array_equal(full[:, :200], top) and np.array_equal(full[:, :5000], big)
    assert np.all(full[2] == -1)
    for r in (0, 1, 3):
        valid = int(np.sum(~np.isnan(s[r])))
        assert np.all(full[r, valid:] == -1) and np.all(full[r, :valid] >= 0)
        assert np.all(np.diff(s[r][full[r, :valid]]) <= 0)


def test_pipeline_recommender_without_n(gpu, ml_small):
    "std:topn without `n` on ml-latest-small (9 125 items > 4096): ranks every candidate"
    from pathlib import Path

    from lkpy_amd.data import load_movielens_npz
    from lkpy_amd.pipeline import Pipeline
    from lkpy_amd.training import TrainingOptions

    golden = Path(__file__).parent / "golden"
    ds = load_movielens_npz(golden / "ml_small.npz")
    pipe = Pipeline.load_config(golden / "pipelines" / "als-implicit.toml")
    pipe.node("scorer").component.config.epochs = 2
    pipe.train(ds, TrainingOptions(rng=42))
    uid = int(ml_small["user_ids"][5])
    recs = pipe.run("recommender", query=uid)  # no n: TopNConfig.n is None -> all
    hist = ds.user_row(uid).ids()
    assert len(recs) == len(ml_small["item_ids"]) - len(hist) > 4096
    assert np.all(np.diff(recs.scores()) <= 0) and not np.isin(recs.ids(), hist).any()


def test_cancel_before_start_and_progress(gpu, oracle, rng):
    "AccelTask.cancel / current_progress reach the kernels through lk_task_ctl"
    from lkpy_amd import _accel
    from lkpy_amd import _device as D
    from lkpy_amd import _native
    from lkpy_amd.parallel import run_accel_task

    n_rows, n_cols, k = 5000, 800, 32
    mat = sps.random_array((n_rows, n_cols), density=0.02, format="csr", dtype=np.float32, rng=rng)
    mat.data[:] = 40.0
    other = (rng.standard_normal((n_cols, k)) * 0.1).astype(np.float32)
    this0 = (rng.standard_normal((n_rows, k)) * 0.1).astype(np.float32)
    otor = oracle.implicit_otor(other, 0.1)

    # (1) normal completion: progress reaches the row count, result as usual
    this = this0.copy()
    task = _accel.als.train_implicit_matrix(mat, this, other, otor)
    run_accel_task(task)
    assert task.current_progress() == (n_rows, n_rows)
    want = this0.copy()
    oracle.als_half_epoch(mat, want, other, otor)
    assert np.linalg.norm(this - want) / np.linalg.norm(want) < 1e-4

    # (2) cancelled before the launch: every row is skipped, LK_E_CANCELLED comes back
    csr = D.DeviceCSR.from_scipy(mat, gpu)
    plan = D.ALSPlan(csr, k, _native.SOLVER_CHOLESKY)
    ctl = D.TaskCtl()
    plan.set_ctl(ctl)
    ctl.cancel()
    d_this = D.to_device_padded(this0, gpu)
    plan.half_epoch(d_this, D.to_device_padded(other, gpu), D.Gramian(k, gpu)(
        D.to_device_padded(other, gpu), 0.1))
    with pytest.raises(KeyboardInterrupt):
        plan.check_status()
    done, total = ctl.progress()
    assert total == n_rows and done < n_rows
    # rows never started are untouched (all of them here: the flag was up before the launch)
    assert np.array_equal(D.to_host_unpadded(d_this, k), this0)
    # the block is reusable after a reset
    ctl.reset()
    plan.half_epoch(d_this, D.to_device_padded(other, gpu), D.Gramian(k, gpu)(
        D.to_device_padded(other, gpu), 0.1))
    plan.check_status()
    assert ctl.progress() == (n_rows, n_rows)

    # (3) through the task protocol: cancel() before invoke -> KeyboardInterrupt inside
    t2 = _accel.als.train_implicit_matrix(mat, this0.copy(), other, otor)
    t2.cancel()
    with pytest.raises(RuntimeError, match="accelerator task failed"):
        run_accel_task(t2)


def test_live_progress_and_cancel_mid_flight(gpu, oracle):
    """A build long enough to be observed: the main thread sees the row count grow while the
    kernel runs and a cancel() issued mid-flight stops it early (tasks/mod.rs:88-105)."""
    import torch

    from lkpy_amd import _device as D
    from lkpy_amd import synth

    mat = synth.ml25m_like(seed=11, scale=0.35)
    dui, diu, _m, _ = D.iknn_prepare(mat, True, gpu)
    n_items = mat.shape[1]

    ctl = D.TaskCtl()
    seen, err = [], []

    def work():
        try:
            D.iknn_build(dui, diu, 1.0e-6, None, ctl=ctl)
        except BaseException as e:  # noqa: BLE001
            err.append(e)

    th = threading.Thread(target=work)
    t0 = time.perf_counter()
    th.start()
    while th.is_alive():
        seen.append(ctl.progress()[0])
        time.sleep(0.0005)
    th.join()
    dt = time.perf_counter() - t0
    assert not err, err
    assert ctl.progress() == (n_items, n_items)
    assert all(b >= a for a, b in zip(seen, seen[1:]))  # monotone
    mid = [s for s in seen if 0 < s < n_items]
    print(f"\nlive progress: {len(seen)} polls in {dt * 1e3:.1f} ms, {len(mid)} mid-flight values")

    # cancel while the kernel runs: returns LK_E_CANCELLED with fewer rows than the total
    ctl2 = D.TaskCtl()
    err2 = []

    def work2():
        try:
            D.iknn_build(dui, diu, 1.0e-6, None, ctl=ctl2)
        except BaseException as e:  # noqa: BLE001
            err2.append(e)

    th2 = threading.Thread(target=work2)
    th2.start()
    while th2.is_alive() and ctl2.progress()[0] == 0:
        time.sleep(0.0002)
    ctl2.cancel()
    th2.join()
    torch.cuda.synchronize()
    done, total = ctl2.progress()
    print(f"cancel mid-flight: stopped at {done} of {total} rows, error: {err2!r}")
    # the race is real (the build may finish first on a fast day); when the cancel won, the
    # call must say so and must have stopped early
    if err2:
        assert isinstance(err2[0], KeyboardInterrupt) and done < total
    else:
        assert done == total
