"""
GPU parity AT BASELINE SCALE (cfg2 / cfg3 shapes): the ML-25M-shaped synthetic that
``bench.py`` measures on, not a shrunken stand-in.

* ALS (cfg2 k = 64, and the same data at k = 128 / 256: the kernels of cfg4 / cfg5): one full
  epoch from a trained state, GPU and oracle run FROM IDENTICAL INPUTS for each half, every one
  of the 162 541 + 62 423 rows compared (k = 256: a 25 % sample + the 64 longest rows): on the
  DEFAULT path (hybrid summation order, round 5) no decidable row may be further than 1e-4 from
  the oracle's (``oracle/parity.py``; src/accel/als/implicit.rs:87-125) -- no allowance, no
  re-run; ``LK_ALS_RHS_ORDER=accurate`` (round 4's default) is run beside it to show the rows
  the hybrid order is there for.
* item-kNN (cfg3): >= 2 000 sampled rows of the 62 423-item build compared BITWISE with the
  oracle's ``sim_row`` (src/accel/knn/item_train.rs:95-152) -- the staged single-pass path
  (staging offsets beyond 2^32) and the two-pass path (``LK_IKNN_STAGE_GB=0``).
"""
import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ml25m():
    from lkpy_amd import synth

    return synth.ml25m_like()


@pytest.mark.parametrize("k,row_frac,epochs", [(64, 1.0, 25), (128, 1.0, 25), (256, 0.25, 20)])
def test_als_epoch_at_scale(gpu, oracle, ml25m, k, row_frac, epochs):
    """One epoch from a trained state at the ML-25M shape, k = 64 (cfg2), 128 (cfg4's kernel) and
    256 (cfg5's kernel): GPU and oracle FROM IDENTICAL INPUTS for each half, every row (k = 256: a
    25 % row sample plus the 64 longest rows -- the oracle's dense sposv per row is 16 x the
    k = 64 cost).  Criterion, on the DEFAULT path: every decidable row (cond * u < 1e-5) within
    the raw 1e-4 of the oracle's row, the others inside the forward bound of a float32 solve
    (``accounted``).  Round 4 allowed up to 64 long rows to be re-run through a reference-order
    plan; the default plan now evaluates those rows in the reference's order itself."""
    import torch

    from lkpy_amd import _native
    from lkpy_amd._als_engine import HipBackend, ImplicitALSEngine
    from oracle import parity

    reg, weight = 0.1, 40.0
    ui = sps.csr_array((np.full(ml25m.nnz, weight, dtype=np.float32), ml25m.indices,
                        ml25m.indptr), shape=ml25m.shape)
    iu = sps.csr_array(ui.T)
    iu.sort_indices()
    rng = np.random.default_rng(42)
    Q0 = oracle.als_initial_params(rng, ui.shape[1], k)
    P0 = oracle.als_initial_params(rng, ui.shape[0], k)
    eng = ImplicitALSEngine(ui, k, reg, reg, P0, Q0, HipBackend(k, gpu, _native.SOLVER_CHOLESKY))
    assert eng.u_plan.order_mode == "auto" and eng.i_plan.long_rows() > 100
    # a TRAINED state: the first epochs after the tiny init are ill-conditioned (cond(A) ~ 4e3
    # after 4 epochs: thousands of rows where the oracle itself is 1e-4 from float64)
    for _ in range(epochs):
        eng.train_epoch()
    eng.check()
    P, Q = eng.user_embeddings(), eng.item_embeddings()
    eng.train_epoch()
    eng.check()
    P1, Q1 = eng.user_embeddings(), eng.item_embeddings()
    # the same epoch in round 4's default order (reported, not asserted)
    acc_eng = ImplicitALSEngine(ui, k, reg, reg, P, Q,
                                HipBackend(k, gpu, _native.SOLVER_CHOLESKY, "accurate"))
    acc_eng.train_epoch()
    acc_eng.check()
    Pa, Qa = acc_eng.user_embeddings(), acc_eng.item_embeddings()
    del acc_eng
    torch.cuda.synchronize()

    srng = np.random.default_rng(5)
    report = {}
    # user half: inputs (P, Q); item half: inputs (Q, P1 as the GPU produced it)
    for name, mat, this, other, got, got_acc in (("user", ui, P, Q, P1, Pa),
                                                 ("item", iu, Q, P1, Q1, None)):
        lens = np.diff(mat.indptr)
        empty = lens == 0
        assert np.all(got[empty] == 0)  # implicit.rs:98-101
        if row_frac < 1.0:
            rows = np.sort(srng.choice(mat.shape[0], int(mat.shape[0] * row_frac), replace=False))
            rows = np.union1d(rows, np.argsort(-lens, kind="stable")[:64])  # + the longest rows
            mat, this, got = sps.csr_array(mat[rows]), this[rows], got[rows]
            got_acc = got_acc[rows] if got_acc is not None else None
        want = np.ascontiguousarray(this.copy())
        otor = oracle.implicit_otor(other, reg)
        oracle.als_half_epoch(mat, want, other, otor)
        exact, cond = oracle.als_referee_f64(mat, other, reg)
        acc = parity.als_half_accounting(got, want, exact, cond)
        num = np.linalg.norm(got.astype(np.float64) - want, axis=1)
        den = np.linalg.norm(want.astype(np.float64), axis=1)
        rel = num / np.maximum(den, 1e-300)
        # rows over 1e-4 where 1e-4 is decidable at all (cond * 2^-24 < 1e-5, oracle/parity.py);
        # the others are covered by `accounted` (forward bound of a float32 solve)
        decidable = cond * parity.U32 < 1.0e-5
        acc["rows_over_all"] = np.flatnonzero((rel > 1e-4) & decidable)
        long_rows = np.diff(mat.indptr) > 2048
        acc["long_rows"] = int(long_rows.sum())
        acc["long_rows_rel_max"] = float(rel[long_rows].max()) if long_rows.any() else 0.0
        if got_acc is not None:  # (user half only: the item half's inputs differ between modes)
            rel_a = np.linalg.norm(got_acc.astype(np.float64) - want, axis=1) / np.maximum(den, 1e-300)
            acc["accurate_mode_long_rows_rel_max"] = (float(rel_a[long_rows].max())
                                                      if long_rows.any() else 0.0)
            acc["accurate_mode_rows_over_5e-5"] = int(((rel_a > 5e-5) & decidable).sum())
        report[name] = acc
    print(f"\nk = {k} at-scale ALS parity ({'every row' if row_frac >= 1 else f'{row_frac:.0%} sample'}):")
    for name, acc in report.items():
        print(" ", name, {k_: v for k_, v in acc.items()
                          if k_ not in ("by_cond_decade", "rows_over_all", "exceptions")})
    for name, acc in report.items():
        # the RAW north-star criterion, every row, nothing folded in: what bench.py prints as
        # ``parity.ok`` (asserted here, not only printed there)
        assert acc["ok"] and acc["rows_over_1e-4"] == 0, (name, acc["rows_over_1e-4"],
                                                            acc["row_rel_max"], acc.get("exceptions"))
        assert acc["accounted"], (name, acc)
        assert len(acc["rows_over_all"]) == 0, (name, acc["rows_over_all"][:10], acc)
        # the rows evaluated in the reference's order sit an order of magnitude inside 1e-4
        assert acc["long_rows_rel_max"] < 3e-5, (name, acc["long_rows_rel_max"])



def test_als_item_half_at_cfg5_shape(gpu, oracle):
    """BASELINE configs[4]'s matrix (10^7 users x 10^6 items x 10^8 entries generated in HBM,
    k = 256), item half from a trained state: EVERY item row of more than 4096 entries (about
    2 000 rows holding 60 % of the entries, the busiest 1.5 M entries long -- the rows where the
    reference's sequential float32 sums drift and the default plan must follow its order) plus
    a seeded 0.25 % sample of the others, GPU vs oracle from identical inputs: no row further
    than the raw 1e-4 (src/accel/als/implicit.rs:110-119).  Until round 6 this lived only in
    bench.py's cfg5 leg."""
    import torch

    from lkpy_amd import _native, synth
    from lkpy_amd._als_engine import HipBackend, ImplicitALSEngine
    from oracle import parity

    k, reg = 256, 0.1
    c = synth.CFG5
    csr = synth.zipf_csr_on_device(gpu, c["n_users"], c["n_items"], c["nnz"], seed=c["seed"],
                                   value=40.0)
    backend = HipBackend(k, gpu, _native.SOLVER_AUTO)
    eng = ImplicitALSEngine(csr, k, reg, reg, None, None, backend)
    del csr
    for _ in range(4):  # (a trained state: the first epochs from the init are ill-conditioned)
        eng.train_epoch()
    eng.check()
    plan = eng.i_plan
    assert plan.order_mode == "auto"
    hp = plan.csr.h_indptr.astype(np.int64)
    lens_all = np.diff(hp)
    long_rows = np.flatnonzero(lens_all > 4096)
    assert len(long_rows) > 1500 and lens_all.max() > 1_000_000
    rng = np.random.default_rng(3)
    rows = np.unique(np.concatenate([rng.choice(len(lens_all), len(lens_all) // 400,
                                                replace=False), long_rows]))
    lens = lens_all[rows]
    ptr = np.zeros(len(rows) + 1, np.int64)
    np.cumsum(lens, out=ptr[1:])
    take = torch.from_numpy(np.concatenate([np.arange(hp[r], hp[r + 1]) for r in rows])).to(gpu)
    sub = sps.csr_array((plan.csr.values[take].cpu().numpy(), plan.csr.indices[take].cpu().numpy(),
                         ptr), shape=(len(rows), eng.P.shape[0]))
    del take
    # the GPU's item half from (Q, P): one more half-epoch, the sampled rows read back
    other_h = backend.download(eng.P)
    otor = backend.gramian(eng.P, reg)
    backend.half_epoch(plan, eng.Q[eng.i_lo:eng.i_hi], eng.P, otor)
    plan.check_status()
    got = backend.download(eng.Q[eng.i_lo:eng.i_hi][torch.from_numpy(rows).to(gpu)])
    del eng
    torch.cuda.empty_cache()
    want = np.zeros_like(got)
    oracle.als_half_epoch(sub, want, other_h, oracle.implicit_otor(other_h, reg))
    # (no float64 referee here: the raw criterion needs none, and on these rows it doubles the
    # test's two minutes of host time; bench.py's cfg5 leg runs it)
    acc = parity.als_half_accounting(got, want, None, None)
    rel = np.linalg.norm(got.astype(np.float64) - want, axis=1) / \
        np.maximum(np.linalg.norm(want.astype(np.float64), axis=1), 1e-300)
    is_long = lens > 4096
    print(f"\ncfg5 item half: {len(rows)} rows checked ({int(is_long.sum())} of more than 4096 "
          f"entries, longest {int(lens.max())}); rows over 1e-4: {acc['rows_over_1e-4']}; worst "
          f"{acc['row_rel_max']:.2e} (long rows {rel[is_long].max():.2e})")
    assert acc["ok"] and acc["rows_over_1e-4"] == 0, (acc["rows_over_1e-4"], acc["row_rel_max"])


def _sample_rows(rng, n_items, n):
    return np.sort(rng.choice(n_items, n, replace=False)).astype(np.int32)


def _gather_rows(out, rows):
    "rows of a DeviceCSR similarity matrix -> host (ptr, idx, val)"
    import torch

    r = torch.as_tensor(rows.astype(np.int64), device=out.indptr.device)
    beg, end = out.indptr[r], out.indptr[r + 1]
    lens = end - beg
    ptr = torch.zeros(len(rows) + 1, dtype=torch.int64, device=lens.device)
    ptr[1:] = torch.cumsum(lens, 0)
    pos = torch.arange(int(ptr[-1].item()), device=lens.device)
    src = pos - torch.repeat_interleave(ptr[:-1], lens) + torch.repeat_interleave(beg, lens)
    return ptr.cpu().numpy(), out.indices[src].cpu().numpy(), out.values[src].cpu().numpy()


@pytest.mark.parametrize("staged", [True, False])
def test_knn_build_cfg3_sampled_rows(gpu, oracle, ml25m, staged, monkeypatch):
    import torch

    from lkpy_amd import _device as D
    from oracle import parity

    if not staged:
        monkeypatch.setenv("LK_IKNN_STAGE_GB", "0")
    dui, diu, _means, _ = D.iknn_prepare(ml25m, True, gpu)
    out = D.iknn_build(dui, diu, 1.0e-6, None)
    assert out.indptr.dtype == torch.int64
    nnz = int(out.indices.shape[0])
    assert nnz > 2**29  # the regime no small test reaches
    # the oracle side consumes the SAME normalised matrices (their bit-identity with the
    # reference's SciPy preparation is tests/test_gpu_iknn_prepare.py's subject)
    ui = sps.csr_array((dui.values.cpu().numpy(), dui.indices.cpu().numpy(), dui.h_indptr),
                       shape=dui.shape)
    iu = sps.csr_array((diu.values.cpu().numpy(), diu.indices.cpu().numpy(), diu.h_indptr),
                       shape=diu.shape)
    rng = np.random.default_rng(3)
    rows = _sample_rows(rng, ui.shape[1], 2048)
    # plus the heaviest rows (longest item columns): the multi-window / long-slice regime
    heavy = np.argsort(-np.diff(iu.indptr))[:16].astype(np.int32)
    rows = np.unique(np.concatenate([rows, heavy]))
    want = oracle.iknn_build_rows(ui, iu, rows, 1.0e-6, None)
    got = _gather_rows(out, rows)
    res = parity.knn_rows_equal(*got, want)
    print("\ncfg3 at-scale kNN parity (%s):" % ("staged" if staged else "two-pass"), res,
          "of", nnz, "similarities")
    assert res["bitwise_equal"], res
    # structural invariants over the WHOLE output (size-independent properties)
    ptr = out.indptr
    assert int(ptr[0].item()) == 0 and int(ptr[-1].item()) == nnz
    assert bool((ptr[1:] >= ptr[:-1]).all())
    assert float(out.values.min().item()) >= 1.0e-6  # item_train.rs:135


def test_topk_cfg2_all_users_with_exclusions(gpu, oracle, ml25m):
    """The fused dense top-N call at its bench shape: ALL 162 541 users x 62 423 items, n = 100,
    every user's own items excluded, factors from real epochs (3 launches x 512 workgroups, LDS
    candidate counters, redo list) -- a seeded sample of users against the oracle's per-query
    path (scores = Q u in the k-ordered chain, candidates minus history, heap top-N) FROM THE
    SAME FACTORS: index sets and score bits identical; the order inside the list may differ only
    among items whose scores are bit-equal (the reference leaves it unspecified)."""
    import torch

    from lkpy_amd import _device as D
    from lkpy_amd import _native
    from lkpy_amd._als_engine import HipBackend, ImplicitALSEngine

    k, reg = 64, 0.1
    ui = sps.csr_array((np.full(ml25m.nnz, 40.0, dtype=np.float32), ml25m.indices, ml25m.indptr),
                       shape=ml25m.shape)
    rng = np.random.default_rng(42)
    Q0 = oracle.als_initial_params(rng, ui.shape[1], k)
    P0 = oracle.als_initial_params(rng, ui.shape[0], k)
    backend = HipBackend(k, gpu, _native.SOLVER_CHOLESKY)
    eng = ImplicitALSEngine(ui, k, reg, reg, P0, Q0, backend)
    for _ in range(4):
        eng.train_epoch()
    eng.check()
    h_ptr = eng.u_plan.csr.h_indptr.astype(np.int64)
    excl_ptr = torch.from_numpy(h_ptr).to(gpu)
    g_idx, g_sc = D.score_topk(eng.P, eng.Q, k, 100, excl_ptr, eng.u_plan.csr.indices)
    torch.cuda.synchronize()
    P, Q = backend.download(eng.P), backend.download(eng.Q)  # the engine's row order, both sides
    ex_idx = eng.u_plan.csr.indices.cpu().numpy()
    users = np.sort(np.random.default_rng(9).choice(P.shape[0], 4096, replace=False))
    # always include the heaviest users (longest exclusion lists: relabelled rows 0..15)
    users = np.unique(np.concatenate([users, np.arange(16)]))
    lens = h_ptr[users + 1] - h_ptr[users]
    ptr = np.zeros(len(users) + 1, np.int64)
    np.cumsum(lens, out=ptr[1:])
    idx = np.concatenate([ex_idx[h_ptr[u]:h_ptr[u + 1]] for u in users])
    want_i, want_s = oracle.score_topn_batch(Q, P[users], 100, ptr, idx)
    got_i, got_s = g_idx.cpu().numpy()[users], g_sc.cpu().numpy()[users]
    assert np.array_equal(got_s.view(np.uint32), want_s.view(np.uint32))  # sorted score rows
    # index lists: identical, except where different items carry the same score bits (the
    # reference heap's order among equal scores -- inside the list or at the cut -- is its sift
    # order; the GPU takes the lower item number): every listed item must really
    # have the listed score
    differ = np.flatnonzero((got_i != want_i).any(axis=1))
    at_cut = 0
    for r in differ:
        sc = oracle.score_dense(Q, P[users[r]])
        assert np.array_equal(sc[got_i[r]].view(np.uint32), got_s[r].view(np.uint32))
        at_cut += int(not np.array_equal(np.sort(got_i[r]), np.sort(want_i[r])))
    for r in range(len(users)):
        assert not np.isin(got_i[r], idx[ptr[r]:ptr[r + 1]]).any()
    print(f"\ncfg2 top-100 with exclusions: {len(users)} users, score rows bit-identical; "
          f"{len(differ)} lists differ among equal-score items ({at_cut} of them at the cut)")
