/*
 * lk_oracle.c -- CPU restatement of the LensKit hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity ORACLE: a plain-C, line-by-line restatement of the
 * reference's Rust accelerator for the implicit-ALS half-epoch, the item-item
 * similarity build, item-kNN scoring and top-N selection.  It is imported only
 * by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never
 * by the product path (lkpy_amd/), which must fail loudly without the HIP
 * library.
 *
 * Pinning (see oracle/README.md): the item-kNN path is pinned against the
 * reference's golden vector tests/models/item-item-preds.csv and the closed
 * forms in tests/models/test_knn_item_item.py; top-N against the Rust unit
 * tests in src/accel/indirect/heap.rs:105-162 and the properties in
 * tests/accel/test_argsort.py.  The implicit-ALS row solve is pinned against
 * the REFERENCE'S OWN Python row functions executed in this container
 * (tests/golden/make_als_fixtures.py -> als_ref_*.npz, checked by
 * tests/test_oracle_pinned.py): the reference holds no golden factors, but its
 * _train_new_row / solve_cholesky / _implicit_otor / initial_params run here.
 * What remains third party is the summation order inside ndarray's `dot`
 * (matrixmultiply 0.3.11, restated from its published algorithm below) and
 * LAPACK sposv (the very function pointer the reference resolves,
 * src/accel/als/solve.rs:47-59).
 *
 * All citations are relative to /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* LAPACK sposv signature, as resolved by src/accel/als/solve.rs:19-28 from
 * scipy.linalg.cython_lapack.__pyx_capi__["sposv"] (src/accel/cython.rs:16-46). */
typedef void (*lko_sposv_fn)(const char *uplo, const int *n, const int *nrhs, float *a,
                             const int *lda, float *b, const int *ldb, int *info);

/* Thread cap: every row solve calls OpenBLAS' sposv concurrently, and SciPy's
 * bundled OpenBLAS has a fixed pool of per-thread work buffers (built for 64
 * threads); more concurrent callers than that crash it.  The reference itself
 * defaults to min(ncpus, 8) worker threads (src/lenskit/schemas/settings.py:182-185). */
#define LKO_MAX_THREADS 32

int lko_num_threads(void)
{
#ifdef _OPENMP
    int n = omp_get_max_threads();
    int cap = LKO_MAX_THREADS;
    const char *e = getenv("LKO_MAX_THREADS"); /* experiments with the cap (tools/oracle_threads.py) */
    if (e && atoi(e) > 0) cap = atoi(e);
    return n > cap ? cap : n;
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* Implicit ALS: src/accel/als/implicit.rs:56-125                              */
/* ------------------------------------------------------------------------- */

/* `mtl.dot(&o_picked)` (implicit.rs:112) / `mt.dot(&o_picked)` (explicit.rs:103): ndarray 0.17.2
 * built WITHOUT its blas feature (Cargo.toml:38) sends an f32 matrix product to
 * matrixmultiply 0.3.11 (Cargo.lock:701-702) `sgemm`.  That crate is not under /root/reference;
 * its published algorithm (Goto-style packing, src/gemm.rs `gemm_loop`; block sizes
 * src/archparam.rs S_MC = 64, S_KC = 256, S_NC = 1024; x86-64 kernels chosen at run time,
 * `fma` feature -> 8x8 `_mm256_fmadd_ps` micro-kernel, src/sgemm_kernel.rs) fixes the order in
 * which the inner dimension -- here the row's entries -- is summed:
 *
 *   for each chunk of KC = 256 entries, in order:
 *       ab = 0;  for j in chunk, in order:  ab = fma(l[j][f], m[j][g], ab)
 *       c[f][g] = (first chunk) ? ab : c[f][g] + ab          (beta = 0 / 1, alpha = 1)
 *
 * (one accumulator per output element per chunk; MC / NC blocking only reorders *which*
 * elements are worked on, not the sums).  lko_gemm_mode 0 keeps the round-1/2 restatement --
 * one unblocked sequential f32 sum with separately rounded multiply and add -- for A/B
 * comparisons (tests/test_oracle_pinned.py reports both against the reference's own Python row
 * solve).  `l` = m[j][f] * v[j] rounded to f32 (the materialised `mtl`); v == NULL: l = m. */
#define LKO_SGEMM_KC 256
static int lko_gemm_mode = 1;
void lko_set_gemm_mode(int mode) { lko_gemm_mode = mode; }
int lko_get_gemm_mode(void) { return lko_gemm_mode; }

static void lko_gram_mtl_m(const float *m, const float *v, int64_t n, int k, float *a, float *ab)
{
    if (lko_gemm_mode == 0) {
        memset(a, 0, sizeof(float) * k * k);
        for (int64_t j = 0; j < n; j++) {
            const float *mj = m + j * k;
            float vj = v ? v[j] : 1.0f;
            for (int f = 0; f < k; f++) {
                float l = v ? mj[f] * vj : mj[f];
                float *af = a + (int64_t)f * k;
                for (int g = 0; g < k; g++) af[g] += l * mj[g];
            }
        }
        return;
    }
    for (int64_t j0 = 0; j0 < n; j0 += LKO_SGEMM_KC) {
        int64_t j1 = j0 + LKO_SGEMM_KC < n ? j0 + LKO_SGEMM_KC : n;
        float *dst = j0 == 0 ? a : ab;
        memset(dst, 0, sizeof(float) * k * k);
        for (int64_t j = j0; j < j1; j++) {
            const float *mj = m + j * k;
            float vj = v ? v[j] : 1.0f;
            for (int f = 0; f < k; f++) {
                float l = v ? mj[f] * vj : mj[f];
                float *df = dst + (int64_t)f * k;
                for (int g = 0; g < k; g++) df[g] = fmaf(l, mj[g], df[g]);
            }
        }
        if (j0 != 0)
            for (int i = 0; i < k * k; i++) a[i] = a[i] + ab[i];
    }
}

/* One row: train_row_solve, implicit.rs:87-125.  Scratch: m (n*k), a (k*k),
 * y (k).  Returns the squared delta, or a negative LAPACK info code -> err. */
static float lko_als_row(lko_sposv_fn sposv, const int64_t *indptr, const int32_t *indices,
                         const float *values, int64_t row, int k, float *row_data,
                         const float *other, const float *otor, float *m, float *vbuf, float *a,
                         float *ab, float *y, int *err)
{
    int64_t sp = indptr[row], ep = indptr[row + 1];
    int64_t n = ep - sp;
    if (n == 0) { /* implicit.rs:98-101: empty row -> zeros, delta 0 */
        for (int f = 0; f < k; f++) row_data[f] = 0.0f;
        return 0.0f;
    }
    /* o_picked = other.select(Axis(0), cols)  (implicit.rs:108) */
    for (int64_t j = 0; j < n; j++) {
        memcpy(m + j * k, other + (int64_t)indices[sp + j] * k, sizeof(float) * k);
        vbuf[j] = values[sp + j];
    }
    /* mtl = mt * vals; mtm = mtl.dot(o_picked)  (implicit.rs:110-112):
     * A[f][g] = sum_j (M[j][f]*v_j) * M[j][g]; the product M*v is rounded to
     * f32 first (it is materialised as `mtl`), accumulation in f32. */
    lko_gram_mtl_m(m, vbuf, n, k, a, ab);
    /* a = otor + mtm (implicit.rs:115) */
    for (int i = 0; i < k * k; i++) a[i] = otor[i] + a[i];
    /* vals += 1; y = mt.dot(vals) (implicit.rs:116-117) */
    memset(y, 0, sizeof(float) * k);
    for (int64_t j = 0; j < n; j++) {
        const float *mj = m + j * k;
        float v1 = vbuf[j] + 1.0f;
        for (int f = 0; f < k; f++) y[f] += mj[f] * v1;
    }
    /* solver.solve: sposv('U', k, 1, a, k, y, k)  (solve.rs:65-107).  Row-major
     * A handed over as column-major: symmetric, so harmless. */
    {
        char uplo = 'U';
        int kk = k, nrhs = 1, info = 0;
        sposv(&uplo, &kk, &nrhs, a, &kk, y, &kk, &info);
        if (info != 0) {
            *err = info;
            return 0.0f;
        }
    }
    /* deltas = soln - row; row.assign(soln); deltas.dot(deltas) (implicit.rs:121-124) */
    float d2 = 0.0f;
    for (int f = 0; f < k; f++) {
        float d = y[f] - row_data[f];
        row_data[f] = y[f];
        d2 += d * d;
    }
    return d2;
}

/* Half-epoch: ImplicitTrainTask::invoke, implicit.rs:56-84.  `this_` is
 * updated in place; *out_frob = sqrt(sum of squared row deltas) in f32.
 * Rows are distributed over OpenMP threads (the reference uses a rayon
 * par_iter; its reduction order is nondeterministic, ours is fixed: per-thread
 * partials in row order, combined in thread order).  Returns 0 or the LAPACK
 * info of the first failing row (the reference raises RuntimeError("ALS solve
 * error: ..."), implicit.rs:79). */
int lko_als_implicit_half_epoch(void *sposv_ptr, const int64_t *indptr, const int32_t *indices,
                                const float *values, int64_t n_rows, int k, float *this_,
                                const float *other, const float *otor, int n_threads,
                                float *out_frob)
{
    lko_sposv_fn sposv = (lko_sposv_fn)sposv_ptr;
    int64_t max_n = 0;
    for (int64_t r = 0; r < n_rows; r++) {
        int64_t n = indptr[r + 1] - indptr[r];
        if (n > max_n) max_n = n;
    }
    int failed = 0;
    double total = 0.0; /* partials are f32 like the reference; the cross-thread
                           combine order is free in the reference (rayon) */
#ifdef _OPENMP
    if (n_threads <= 0 || n_threads > LKO_MAX_THREADS) {
        int cap = lko_num_threads();
        n_threads = (n_threads <= 0 || n_threads > cap) ? cap : n_threads;
    }
#else
    n_threads = 1;
#endif
    float *partials = (float *)calloc((size_t)n_threads, sizeof(float));
#pragma omp parallel num_threads(n_threads)
    {
#ifdef _OPENMP
        int tid = omp_get_thread_num();
#else
        int tid = 0;
#endif
        float *m = (float *)malloc(sizeof(float) * (size_t)(max_n > 0 ? max_n : 1) * k);
        float *vb = (float *)malloc(sizeof(float) * (size_t)(max_n > 0 ? max_n : 1));
        float *a = (float *)malloc(sizeof(float) * (size_t)k * k);
        float *ab = (float *)malloc(sizeof(float) * (size_t)k * k);
        float *y = (float *)malloc(sizeof(float) * (size_t)k);
        float acc = 0.0f;
#pragma omp for schedule(dynamic, 64)
        for (int64_t r = 0; r < n_rows; r++) {
            int err = 0;
            float d2 = lko_als_row(sposv, indptr, indices, values, r, k, this_ + r * k, other,
                                   otor, m, vb, a, ab, y, &err);
            if (err) {
#pragma omp critical
                if (!failed) failed = err;
            }
            acc += d2;
        }
        partials[tid] = acc;
        free(m);
        free(vb);
        free(a);
        free(ab);
        free(y);
    }
    float frob = 0.0f;
    for (int t = 0; t < n_threads; t++) frob += partials[t];
    (void)total;
    free(partials);
    *out_frob = sqrtf(frob);
    return failed;
}

/* ------------------------------------------------------------------------- */
/* Explicit (biased-MF) ALS: src/accel/als/explicit.rs:56-119                    */
/* ------------------------------------------------------------------------- */

/* One row: train_row_solve, explicit.rs:80-119.  A = M^T M + reg*n*I (the
 * diagonal term is added AFTER the product, explicit.rs:104-107), rhs = M^T vals. */
static float lko_als_explicit_row(lko_sposv_fn sposv, const int64_t *indptr,
                                  const int32_t *indices, const float *values, int64_t row, int k,
                                  float *row_data, const float *other, float reg, float *m,
                                  float *a, float *ab, float *y, int *err)
{
    int64_t sp = indptr[row], ep = indptr[row + 1];
    int64_t n = ep - sp;
    if (n == 0) { /* explicit.rs:91-94 */
        for (int f = 0; f < k; f++) row_data[f] = 0.0f;
        return 0.0f;
    }
    for (int64_t j = 0; j < n; j++)
        memcpy(m + j * k, other + (int64_t)indices[sp + j] * k, sizeof(float) * k);
    lko_gram_mtl_m(m, NULL, n, k, a, ab); /* mtm = mt.dot(&o_picked)  (explicit.rs:103) */
    {
        float dg = reg * (float)n; /* reg * cols.len() as f32 */
        for (int f = 0; f < k; f++) a[(int64_t)f * k + f] += dg;
    }
    memset(y, 0, sizeof(float) * k); /* v = mt.dot(&vals)  (explicit.rs:109) */
    for (int64_t j = 0; j < n; j++) {
        const float *mj = m + j * k;
        float v = values[sp + j];
        for (int f = 0; f < k; f++) y[f] += mj[f] * v;
    }
    {
        char uplo = 'U';
        int kk = k, nrhs = 1, info = 0;
        sposv(&uplo, &kk, &nrhs, a, &kk, y, &kk, &info);
        if (info != 0) {
            *err = info;
            return 0.0f;
        }
    }
    float d2 = 0.0f;
    for (int f = 0; f < k; f++) {
        float d = y[f] - row_data[f];
        row_data[f] = y[f];
        d2 += d * d;
    }
    return d2;
}

/* Half-epoch: ExplicitTrainTask::invoke, explicit.rs:56-77; same contract as the implicit
 * half-epoch above. */
int lko_als_explicit_half_epoch(void *sposv_ptr, const int64_t *indptr, const int32_t *indices,
                                const float *values, int64_t n_rows, int k, float *this_,
                                const float *other, float reg, int n_threads, float *out_frob)
{
    lko_sposv_fn sposv = (lko_sposv_fn)sposv_ptr;
    int64_t max_n = 0;
    for (int64_t r = 0; r < n_rows; r++) {
        int64_t n = indptr[r + 1] - indptr[r];
        if (n > max_n) max_n = n;
    }
    int failed = 0;
#ifdef _OPENMP
    if (n_threads <= 0 || n_threads > LKO_MAX_THREADS) {
        int cap = lko_num_threads();
        n_threads = (n_threads <= 0 || n_threads > cap) ? cap : n_threads;
    }
#else
    n_threads = 1;
#endif
    float *partials = (float *)calloc((size_t)n_threads, sizeof(float));
#pragma omp parallel num_threads(n_threads)
    {
#ifdef _OPENMP
        int tid = omp_get_thread_num();
#else
        int tid = 0;
#endif
        float *m = (float *)malloc(sizeof(float) * (size_t)(max_n > 0 ? max_n : 1) * k);
        float *a = (float *)malloc(sizeof(float) * (size_t)k * k);
        float *ab = (float *)malloc(sizeof(float) * (size_t)k * k);
        float *y = (float *)malloc(sizeof(float) * (size_t)k);
        float acc = 0.0f;
#pragma omp for schedule(dynamic, 64)
        for (int64_t r = 0; r < n_rows; r++) {
            int err = 0;
            float d2 = lko_als_explicit_row(sposv, indptr, indices, values, r, k, this_ + r * k,
                                            other, reg, m, a, ab, y, &err);
            if (err) {
#pragma omp critical
                if (!failed) failed = err;
            }
            acc += d2;
        }
        partials[tid] = acc;
        free(m);
        free(a);
        free(ab);
        free(y);
    }
    float frob = 0.0f;
    for (int t = 0; t < n_threads; t++) frob += partials[t];
    free(partials);
    *out_frob = sqrtf(frob);
    return failed;
}

/* ------------------------------------------------------------------------- */
/* REFEREE (not the reference): the implicit half-epoch in float64 + cond(A)    */
/* ------------------------------------------------------------------------- */

/* The exact answer both float32 implementations (the reference's ndarray + sposv path
 * restated above, and the HIP kernels) approximate: OtOr, every normal matrix and every solve
 * in float64.  Also returns an estimate of cond_2(A) per row (power iteration for the largest
 * eigenvalue, inverse iteration through the float64 Cholesky factor for the smallest; both
 * converge from below, so the estimate is a LOWER bound of the true condition number).  Used
 * by the at-scale parity accounting: a float32 solve can only be expected within
 * ~cond(A)*2^-24 of exact, so "1e-4 vs the reference" is checked on rows whose cond allows it
 * and the others are listed.  otor64: k x k float64 (O^T O + reg I computed by the caller in
 * float64); out_x: n_rows x k float64; out_cond: n_rows (0 for empty rows). */
int lko_als_implicit_referee_f64(const int64_t *indptr, const int32_t *indices,
                                 const float *values, int64_t n_rows, int k, const float *other,
                                 const double *otor64, int n_threads, double *out_x,
                                 double *out_cond)
{
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
    if (n_threads > 256) n_threads = 256;
#else
    n_threads = 1;
#endif
    int bad = 0;
#pragma omp parallel num_threads(n_threads)
    {
        double *a = (double *)malloc(sizeof(double) * (size_t)k * k);
        double *l = (double *)malloc(sizeof(double) * (size_t)k * k);
        double *y = (double *)malloc(sizeof(double) * (size_t)k);
        double *w = (double *)malloc(sizeof(double) * (size_t)k);
        double *t = (double *)malloc(sizeof(double) * (size_t)k);
#pragma omp for schedule(dynamic, 16)
        for (int64_t r = 0; r < n_rows; r++) {
            int64_t sp = indptr[r], ep = indptr[r + 1];
            double *x = out_x + r * k;
            if (ep == sp) {
                for (int f = 0; f < k; f++) x[f] = 0.0;
                if (out_cond) out_cond[r] = 0.0;
                continue;
            }
            memcpy(a, otor64, sizeof(double) * (size_t)k * k);
            memset(y, 0, sizeof(double) * (size_t)k);
            for (int64_t j = sp; j < ep; j++) {
                const float *q = other + (int64_t)indices[j] * k;
                double v = (double)values[j];
                for (int f = 0; f < k; f++) {
                    double lf = (double)q[f] * v;
                    double *af = a + (int64_t)f * k;
                    for (int g = 0; g < k; g++) af[g] += lf * (double)q[g];
                    y[f] += (double)q[f] * (v + 1.0);
                }
            }
            /* Cholesky A = L L^T (lower, row-major) */
            int ok = 1;
            memcpy(l, a, sizeof(double) * (size_t)k * k);
            for (int j = 0; j < k && ok; j++) {
                double d = l[j * k + j];
                for (int c = 0; c < j; c++) d -= l[j * k + c] * l[j * k + c];
                if (!(d > 0.0)) {
                    ok = 0;
                    break;
                }
                d = sqrt(d);
                l[j * k + j] = d;
                for (int i = j + 1; i < k; i++) {
                    double s = l[i * k + j];
                    for (int c = 0; c < j; c++) s -= l[i * k + c] * l[j * k + c];
                    l[i * k + j] = s / d;
                }
            }
            if (!ok) {
#pragma omp atomic write
                bad = 1;
                continue;
            }
#define LKO_SOLVE(vec)                                                    \
    do {                                                                  \
        for (int i = 0; i < k; i++) {                                     \
            double s = (vec)[i];                                          \
            for (int c = 0; c < i; c++) s -= l[i * k + c] * (vec)[c];     \
            (vec)[i] = s / l[i * k + i];                                  \
        }                                                                 \
        for (int i = k - 1; i >= 0; i--) {                                \
            double s = (vec)[i];                                          \
            for (int c = i + 1; c < k; c++) s -= l[c * k + i] * (vec)[c]; \
            (vec)[i] = s / l[i * k + i];                                  \
        }                                                                 \
    } while (0)
            memcpy(x, y, sizeof(double) * (size_t)k);
            LKO_SOLVE(x);
            if (out_cond) {
                /* largest eigenvalue: power iteration on A */
                double lmax = 0.0, lmin_inv = 0.0;
                for (int f = 0; f < k; f++) w[f] = 1.0 + 0.37 * (double)((f * 7919) % 13);
                for (int it = 0; it < 30; it++) {
                    double nrm = 0.0;
                    for (int f = 0; f < k; f++) {
                        double s = 0.0;
                        for (int g = 0; g < k; g++) s += a[f * k + g] * w[g];
                        t[f] = s;
                        nrm += s * s;
                    }
                    nrm = sqrt(nrm);
                    lmax = nrm; /* ||A w|| with ||w|| = 1 */
                    for (int f = 0; f < k; f++) w[f] = t[f] / nrm;
                    if (it == 0) lmax = 0.0;
                }
                /* smallest eigenvalue: inverse iteration (A^-1 through the factor) */
                for (int f = 0; f < k; f++) w[f] = 1.0 + 0.61 * (double)((f * 104729) % 11);
                {
                    double n0 = 0.0;
                    for (int f = 0; f < k; f++) n0 += w[f] * w[f];
                    n0 = sqrt(n0);
                    for (int f = 0; f < k; f++) w[f] /= n0;
                }
                for (int it = 0; it < 30; it++) {
                    LKO_SOLVE(w);
                    double nrm = 0.0;
                    for (int f = 0; f < k; f++) nrm += w[f] * w[f];
                    nrm = sqrt(nrm);
                    lmin_inv = nrm;
                    for (int f = 0; f < k; f++) w[f] /= nrm;
                }
                out_cond[r] = lmax * lmin_inv;
            }
#undef LKO_SOLVE
        }
        free(a);
        free(l);
        free(y);
        free(w);
        free(t);
    }
    return bad;
}

/* ------------------------------------------------------------------------- */
/* Item-item similarity build: src/accel/knn/item_train.rs:95-152              */
/* ------------------------------------------------------------------------- */

typedef struct {
    int32_t idx;
    float sim;
} lko_pair;

/* stable merge sort on lko_pair with a comparator (Rust's sort_by_key is a
 * stable sort: item_train.rs:144,149). */
static void lko_msort(lko_pair *a, lko_pair *tmp, int64_t n, int by_sim_desc)
{
    if (n < 2) return;
    int64_t h = n / 2;
    lko_msort(a, tmp, h, by_sim_desc);
    lko_msort(a + h, tmp, n - h, by_sim_desc);
    int64_t i = 0, j = h, o = 0;
    while (i < h && j < n) {
        int take_right;
        if (by_sim_desc)
            take_right = a[j].sim > a[i].sim; /* Reverse(NotNan): larger first; ties keep left */
        else
            take_right = a[j].idx < a[i].idx;
        tmp[o++] = take_right ? a[j++] : a[i++];
    }
    while (i < h) tmp[o++] = a[i++];
    while (j < n) tmp[o++] = a[j++];
    memcpy(a, tmp, sizeof(lko_pair) * (size_t)n);
}

/* sim_row (item_train.rs:95-152) into caller scratch; returns the number of
 * kept neighbours, written to `out` sorted by column. */
static int64_t lko_sim_row(int64_t row, const int64_t *ui_ptr, const int32_t *ui_idx,
                           const float *ui_val, const int64_t *iu_ptr, const int32_t *iu_idx,
                           const float *iu_val, float min_sim, int64_t save_nbrs, int32_t *counts,
                           float *dots, int32_t *used, lko_pair *out, lko_pair *tmp)
{
    int64_t n_used = 0;
    for (int64_t i = iu_ptr[row]; i < iu_ptr[row + 1]; i++) {
        int32_t u = iu_idx[i];
        float r = iu_val[i];
        for (int64_t j = ui_ptr[u]; j < ui_ptr[u + 1]; j++) {
            int32_t other = ui_idx[j];
            if (other == row) continue; /* item_train.rs:120-122 */
            float orate = ui_val[j];
            if (counts[other] == 0) used[n_used++] = other;
            counts[other] += 1;
            /* `dots[other] += r * orate` -- Rust does not contract to FMA: the
             * product is rounded, then added (item_train.rs:128).  Keep it so. */
            volatile float prod = r * orate;
            dots[other] = dots[other] + prod;
        }
    }
    int64_t n = 0;
    for (int64_t q = 0; q < n_used; q++) {
        int32_t i = used[q];
        if (dots[i] >= min_sim) { /* item_train.rs:135 */
            out[n].idx = i;
            out[n].sim = dots[i];
            n++;
        }
        counts[i] = 0; /* reset scratch (the reference allocates fresh vectors per row) */
        dots[i] = 0.0f;
    }
    if (save_nbrs > 0) { /* item_train.rs:140-147 */
        lko_msort(out, tmp, n, 1);
        if (n > save_nbrs) n = save_nbrs;
    }
    lko_msort(out, tmp, n, 0); /* item_train.rs:149 */
    return n;
}

/* compute_similarities (item_train.rs:33-93) + ArrowCSRConsumer
 * (src/accel/sparse/consumer.rs:24-142, order-preserving).  Output is a CSR
 * with int64 offsets (LargeList), malloc'ed here; free with lko_free. */
static int lko_iknn_build_sel(const int64_t *ui_ptr, const int32_t *ui_idx, const float *ui_val,
                              const int64_t *iu_ptr, const int32_t *iu_idx, const float *iu_val,
                              int64_t n_items, const int32_t *sel, int64_t n_out, float min_sim,
                              int64_t save_nbrs, int n_threads, int64_t *out_ptr,
                              int32_t **out_idx, float **out_val)
{
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
    if (n_threads > 256) n_threads = 256;
#else
    n_threads = 1;
#endif
    lko_pair **rows = (lko_pair **)calloc((size_t)(n_out > 0 ? n_out : 1), sizeof(lko_pair *));
    int64_t *lens = (int64_t *)calloc((size_t)n_out + 1, sizeof(int64_t));
#pragma omp parallel num_threads(n_threads)
    {
        int32_t *counts = (int32_t *)calloc((size_t)n_items, sizeof(int32_t));
        float *dots = (float *)calloc((size_t)n_items, sizeof(float));
        int32_t *used = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_items);
        lko_pair *out = (lko_pair *)malloc(sizeof(lko_pair) * (size_t)n_items);
        lko_pair *tmp = (lko_pair *)malloc(sizeof(lko_pair) * (size_t)n_items);
#pragma omp for schedule(dynamic, 16)
        for (int64_t r = 0; r < n_out; r++) {
            int64_t n = lko_sim_row(sel ? (int64_t)sel[r] : r, ui_ptr, ui_idx, ui_val, iu_ptr, iu_idx, iu_val, min_sim,
                                    save_nbrs, counts, dots, used, out, tmp);
            lens[r] = n;
            if (n > 0) {
                rows[r] = (lko_pair *)malloc(sizeof(lko_pair) * (size_t)n);
                memcpy(rows[r], out, sizeof(lko_pair) * (size_t)n);
            }
        }
        free(counts);
        free(dots);
        free(used);
        free(out);
        free(tmp);
    }
    int64_t total = 0;
    for (int64_t r = 0; r < n_out; r++) {
        out_ptr[r] = total;
        total += lens[r];
    }
    out_ptr[n_out] = total;
    int32_t *oi = (int32_t *)malloc(sizeof(int32_t) * (size_t)(total > 0 ? total : 1));
    float *ov = (float *)malloc(sizeof(float) * (size_t)(total > 0 ? total : 1));
#pragma omp parallel for schedule(static) num_threads(n_threads)
    for (int64_t r = 0; r < n_out; r++) {
        int64_t base = out_ptr[r];
        for (int64_t q = 0; q < lens[r]; q++) {
            oi[base + q] = rows[r][q].idx;
            ov[base + q] = rows[r][q].sim;
        }
        free(rows[r]);
    }
    free(rows);
    free(lens);
    *out_idx = oi;
    *out_val = ov;
    return 0;
}

int lko_iknn_build(const int64_t *ui_ptr, const int32_t *ui_idx, const float *ui_val,
                   const int64_t *iu_ptr, const int32_t *iu_idx, const float *iu_val,
                   int64_t n_users, int64_t n_items, float min_sim, int64_t save_nbrs,
                   int n_threads, int64_t *out_ptr, int32_t **out_idx, float **out_val)
{
    (void)n_users;
    return lko_iknn_build_sel(ui_ptr, ui_idx, ui_val, iu_ptr, iu_idx, iu_val, n_items, NULL,
                              n_items, min_sim, save_nbrs, n_threads, out_ptr, out_idx, out_val);
}

/* The same for a SUBSET of output rows (`rows[n_sel]`, any order): out_ptr has n_sel + 1
 * entries, row q of the result is sim_row(rows[q]).  Used for at-scale parity checks where
 * the full matrix (~10^9 entries on ML-25M) is too large to build on the host. */
int lko_iknn_build_rows(const int64_t *ui_ptr, const int32_t *ui_idx, const float *ui_val,
                        const int64_t *iu_ptr, const int32_t *iu_idx, const float *iu_val,
                        int64_t n_items, const int32_t *rows, int64_t n_sel, float min_sim,
                        int64_t save_nbrs, int n_threads, int64_t *out_ptr, int32_t **out_idx,
                        float **out_val)
{
    return lko_iknn_build_sel(ui_ptr, ui_idx, ui_val, iu_ptr, iu_idx, iu_val, n_items, rows,
                              n_sel, min_sim, save_nbrs, n_threads, out_ptr, out_idx, out_val);
}

/* sim_row for a SUBSET of rows, results discarded except the kept-neighbour count:
 * used to time a bounded sample of the build (bench.py cpu_baseline). */
int64_t lko_iknn_sample_rows(const int64_t *ui_ptr, const int32_t *ui_idx, const float *ui_val,
                             const int64_t *iu_ptr, const int32_t *iu_idx, const float *iu_val,
                             int64_t n_items, const int32_t *rows, int64_t n_sel, float min_sim,
                             int64_t save_nbrs, int n_threads)
{
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
    if (n_threads > 256) n_threads = 256;
#else
    n_threads = 1;
#endif
    int64_t kept = 0;
#pragma omp parallel num_threads(n_threads) reduction(+ : kept)
    {
        int32_t *counts = (int32_t *)calloc((size_t)n_items, sizeof(int32_t));
        float *dots = (float *)calloc((size_t)n_items, sizeof(float));
        int32_t *used = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_items);
        lko_pair *out = (lko_pair *)malloc(sizeof(lko_pair) * (size_t)n_items);
        lko_pair *tmp = (lko_pair *)malloc(sizeof(lko_pair) * (size_t)n_items);
#pragma omp for schedule(dynamic, 4)
        for (int64_t q = 0; q < n_sel; q++)
            kept += lko_sim_row(rows[q], ui_ptr, ui_idx, ui_val, iu_ptr, iu_idx, iu_val, min_sim,
                                save_nbrs, counts, dots, used, out, tmp);
        free(counts);
        free(dots);
        free(used);
        free(out);
        free(tmp);
    }
    return kept;
}

void lko_free(void *p) { free(p); }

/* ------------------------------------------------------------------------- */
/* Item-kNN scoring: src/accel/knn/item_score.rs:23-111, knn/accum.rs          */
/* ------------------------------------------------------------------------- */

typedef struct {
    float weight;
    float data;
} lko_acc_entry;

/* ScoreAccumulator (accum.rs:16-22): state 0 Disabled, 1 Empty, 2 Partial(vec),
 * 3 Full(heap).  `e` has capacity limit+1.  The Full state is Rust's
 * std::collections::BinaryHeap on AccEntry with REVERSED ordering (accum.rs:170-
 * 184), i.e. a min-heap on weight stored as a max-heap of reversed keys; we
 * restate std's push (sift_up) and pop (sift_down_to_bottom + sift_up) so the
 * internal array order -- which fixes the f32 summation order of
 * heap.iter().sum() (accum.rs:121-140) -- is the same. */
typedef struct {
    int state;
    int len;
    lko_acc_entry *e;
} lko_acc;

/* "greater" in the heap's (reversed) order == smaller weight */
static inline int lko_hgt(const lko_acc_entry *a, const lko_acc_entry *b)
{
    return a->weight < b->weight;
}
static inline int lko_hle(const lko_acc_entry *a, const lko_acc_entry *b)
{
    /* a <= b in reversed order  <=>  a.weight >= b.weight */
    return a->weight >= b->weight;
}

static void lko_heap_sift_up(lko_acc_entry *d, int start, int pos)
{
    lko_acc_entry elt = d[pos];
    while (pos > start) {
        int parent = (pos - 1) / 2;
        if (lko_hle(&elt, &d[parent])) break; /* hole.element() <= hole.get(parent) */
        d[pos] = d[parent];
        pos = parent;
    }
    d[pos] = elt;
}

static void lko_heap_push(lko_acc *h, lko_acc_entry x)
{
    int old = h->len;
    h->e[h->len++] = x;
    lko_heap_sift_up(h->e, 0, old);
}

static void lko_heap_pop(lko_acc *h)
{
    /* std BinaryHeap::pop: item = data.pop(); if !empty { swap(item, data[0]);
     * sift_down_to_bottom(0) } */
    lko_acc_entry item = h->e[--h->len];
    if (h->len > 0) {
        lko_acc_entry *d = h->e;
        d[0] = item; /* the old root is discarded */
        int end = h->len, start = 0;
        lko_acc_entry elt = d[0];
        int pos = 0, child = 1;
        int limit = end >= 2 ? end - 2 : 0;
        while (child <= limit && end >= 2) {
            /* child += (hole.get(child) <= hole.get(child + 1)) */
            if (lko_hle(&d[child], &d[child + 1])) child += 1;
            d[pos] = d[child];
            pos = child;
            child = 2 * pos + 1;
        }
        if (child == end - 1) {
            d[pos] = d[child];
            pos = child;
        }
        d[pos] = elt;
        lko_heap_sift_up(d, start, pos);
    }
}

/* add_value (accum.rs:100-117) */
static void lko_acc_add(lko_acc *a, int limit, float weight, float value)
{
    if (a->state == 0) return;
    lko_acc_entry x = {weight, value};
    if (a->state == 1) { /* Empty -> Partial(vec) (vector_mut, accum.rs:86-98) */
        a->state = 2;
        a->len = 0;
    }
    if (a->state == 2 && a->len < limit) {
        a->e[a->len++] = x;
        return;
    }
    if (a->state == 2) { /* Partial (full) -> heap: pop from the back, push (accum.rs:76-83) */
        int n = a->len;
        lko_acc_entry *tmp = (lko_acc_entry *)malloc(sizeof(lko_acc_entry) * (size_t)(n + 1));
        memcpy(tmp, a->e, sizeof(lko_acc_entry) * (size_t)n);
        a->len = 0;
        for (int i = n - 1; i >= 0; i--) lko_heap_push(a, tmp[i]);
        free(tmp);
        a->state = 3;
    }
    /* Full: entry.weight > heap.peek().weight  (accum.rs:108) */
    if (a->len > 0 && x.weight > a->e[0].weight) {
        lko_heap_push(a, x);
        while (a->len > limit) lko_heap_pop(a);
    } else if (a->len == 0) { /* limit == 0 cannot happen (PositiveInt) */
        lko_heap_push(a, x);
    }
}

/* score_explicit / score_implicit (item_score.rs:23-111) for ONE query.
 * ref_items may contain negative entries (unknown history items): the reference
 * reads the raw value buffer (item_score.rs:38-49) and would index out of
 * bounds; the restatement SKIPS them.
 * tgt_items < 0 means null target -> null score/count.
 * out_valid[t] = 0 encodes a null score.  explicit != 0 -> weighted average.
 * Returns 0, or 1 if a similarity is NaN (accum.rs:146-151 ValueError). */
int lko_iknn_score(const int64_t *s_ptr, const int32_t *s_idx, const float *s_val,
                   int64_t n_items, const int32_t *ref_items, const float *ref_rates,
                   int64_t n_ref, const int32_t *tgt_items, int64_t n_tgt, int max_nbrs,
                   int min_nbrs, int explicit_, float *out_scores, uint8_t *out_valid,
                   int32_t *out_counts)
{
    lko_acc *accs = (lko_acc *)calloc((size_t)n_items, sizeof(lko_acc));
    for (int64_t t = 0; t < n_tgt; t++) {
        int32_t ti = tgt_items[t];
        if (ti >= 0 && accs[ti].state == 0) {
            accs[ti].state = 1;
            accs[ti].e = (lko_acc_entry *)malloc(sizeof(lko_acc_entry) * (size_t)(max_nbrs + 2));
        }
    }
    int bad = 0;
    for (int64_t q = 0; q < n_ref && !bad; q++) {
        int32_t ri = ref_items[q];
        if (ri < 0 || ri >= n_items) continue;
        float rv = explicit_ ? ref_rates[q] : 0.0f;
        for (int64_t i = s_ptr[ri]; i < s_ptr[ri + 1]; i++) {
            int32_t ti = s_idx[i];
            float sim = s_val[i];
            if (accs[ti].state != 0) {
                if (isnan(sim)) {
                    bad = 1;
                    break;
                }
                lko_acc_add(&accs[ti], max_nbrs, sim, rv);
            }
        }
    }
    for (int64_t t = 0; t < n_tgt; t++) {
        int32_t ti = tgt_items[t];
        out_scores[t] = 0.0f;
        out_valid[t] = 0;
        out_counts[t] = -1;
        if (ti < 0) continue;
        lko_acc *a = &accs[ti];
        int len = (a->state >= 2) ? a->len : 0;
        out_counts[t] = len; /* collect_items_counts, accum.rs:186-199 */
        if (len >= min_nbrs) {
            float tw = 0.0f, ws = 0.0f;
            for (int i = 0; i < len; i++) tw += a->e[i].weight;               /* total_weight */
            for (int i = 0; i < len; i++) ws += a->e[i].weight * a->e[i].data; /* weighted_sum */
            out_scores[t] = explicit_ ? ws / tw : tw; /* accum.rs:209, 231 */
            out_valid[t] = 1;
        }
    }
    for (int64_t i = 0; i < n_items; i++)
        if (accs[i].e) free(accs[i].e);
    free(accs);
    return bad;
}

/* ------------------------------------------------------------------------- */
/* User-kNN scoring: src/accel/knn/user_score.rs:21-98 (same ScoreAccumulator)  */
/* ------------------------------------------------------------------------- */

/* user_score_items_explicit / _implicit: the neighbours (rows of the ratings matrix) are
 * walked in the given order, each pushing (weight = its similarity, value = its rating) into
 * the accumulators of the items it rated; null neighbours (negative row or NaN similarity are
 * the caller's nulls) are skipped (user_score.rs:41-44,80-83).  rat_val NULL = implicit.
 * Returns 1 for a NaN weight reaching an accumulator (accum.rs:146-151). */
int lko_uknn_score(const int64_t *rat_ptr, const int32_t *rat_idx, const float *rat_val,
                   int64_t n_items, const int32_t *nbr_rows, const float *nbr_sims,
                   const uint8_t *nbr_valid, int64_t n_nbrs, const int32_t *tgt_items,
                   int64_t n_tgt, int max_nbrs, int min_nbrs, float *out_scores,
                   uint8_t *out_valid)
{
    const int explicit_ = rat_val != NULL;
    lko_acc *accs = (lko_acc *)calloc((size_t)n_items, sizeof(lko_acc));
    for (int64_t t = 0; t < n_tgt; t++) {
        int32_t ti = tgt_items[t];
        if (ti >= 0 && accs[ti].state == 0) {
            accs[ti].state = 1;
            accs[ti].e = (lko_acc_entry *)malloc(sizeof(lko_acc_entry) * (size_t)(max_nbrs + 2));
        }
    }
    int bad = 0;
    for (int64_t q = 0; q < n_nbrs && !bad; q++) {
        if (nbr_valid && !nbr_valid[q]) continue;
        int32_t nb = nbr_rows[q];
        float sim = nbr_sims[q];
        for (int64_t i = rat_ptr[nb]; i < rat_ptr[nb + 1]; i++) {
            int32_t item = rat_idx[i];
            if (accs[item].state != 0) {
                if (isnan(sim)) {
                    bad = 1;
                    break;
                }
                lko_acc_add(&accs[item], max_nbrs, sim, explicit_ ? rat_val[i] : 0.0f);
            }
        }
    }
    for (int64_t t = 0; t < n_tgt; t++) {
        int32_t ti = tgt_items[t];
        out_scores[t] = 0.0f;
        out_valid[t] = 0;
        if (ti < 0) continue;
        lko_acc *a = &accs[ti];
        int len = (a->state >= 2) ? a->len : 0;
        if (len >= min_nbrs) {
            float tw = 0.0f, ws = 0.0f;
            for (int i = 0; i < len; i++) tw += a->e[i].weight;
            for (int i = 0; i < len; i++) ws += a->e[i].weight * a->e[i].data;
            out_scores[t] = explicit_ ? ws / tw : tw;
            out_valid[t] = 1;
        }
    }
    for (int64_t i = 0; i < n_items; i++)
        if (accs[i].e) free(accs[i].e);
    free(accs);
    return bad;
}

/* ------------------------------------------------------------------------- */
/* Top-N: src/accel/data/sorting.rs:132-172 + src/accel/indirect/heap.rs       */
/* ------------------------------------------------------------------------- */

static void lko_downheap(int32_t *keys, const float *s, int64_t pos, int64_t lim)
{
    /* heap.rs:66-90 (recursive there, iterative here) */
    for (;;) {
        int64_t mn = pos;
        float mv = s[keys[mn]];
        int64_t left = 2 * pos + 1, right = 2 * pos + 2;
        if (left < lim) {
            float lv = s[keys[left]];
            if (lv < mv) {
                mn = left;
                mv = lv;
            }
        }
        if (right < lim) {
            float rv = s[keys[right]];
            if (rv < mv) mn = right;
        }
        if (mn == pos) return;
        int32_t t = keys[pos];
        keys[pos] = keys[mn];
        keys[mn] = t;
        pos = mn;
    }
}

static void lko_upheap(int32_t *keys, const float *s, int64_t pos)
{
    /* heap.rs:92-102 */
    while (pos > 0) {
        int64_t parent = (pos - 1) / 2;
        if (s[keys[parent]] > s[keys[pos]]) {
            int32_t t = keys[pos];
            keys[pos] = keys[parent];
            keys[parent] = t;
            pos = parent;
        } else
            return;
    }
}

/* argtopn_impl (sorting.rs:153-172): `valid` may be NULL (all valid); NaN
 * scores are rejected (sorting.rs:143).  out has room for min(n, len); returns
 * the number written, indices sorted by score descending (heap.rs:56-64). */
int64_t lko_argtopn(const float *scores, const uint8_t *valid, int64_t len, int64_t n,
                    int32_t *out)
{
    if (n <= 0) return 0;
    if (n > len) n = len;
    int64_t size = 0;
    for (int64_t i = 0; i < len; i++) {
        if (valid && !valid[i]) continue;
        if (isnan(scores[i])) continue;
        if (size < n) { /* heap.rs:40-44 */
            out[size] = (int32_t)i;
            lko_upheap(out, scores, size);
            size++;
        } else if (scores[i] > scores[out[0]]) { /* heap.rs:46-51: strictly greater */
            out[0] = (int32_t)i;
            lko_downheap(out, scores, 0, n);
        }
    }
    /* topn_vec, heap.rs:56-64 */
    int64_t m = size;
    while (m > 0) {
        m -= 1;
        int32_t t = out[0];
        out[0] = out[m];
        out[m] = t;
        lko_downheap(out, scores, 0, m);
    }
    return size;
}

/* ------------------------------------------------------------------------- */
/* Dense scoring in a FIXED k-ordered fmaf chain                                */
/* ------------------------------------------------------------------------- */

/* scores[i] = fma(q[i][k-1], u[k-1], ... fma(q[i][0], u[0], 0)).  The
 * reference scores with a NumPy GEMV (src/lenskit/als/_common.py:163-170) whose
 * summation order is BLAS-internal (unpinned); the batched GPU kernel uses the
 * f32 MFMA, which is bit-for-bit this k-ordered chain, so integer top-K index
 * sets can be compared exactly (SURVEY.md section 7 "fp32 summation order"). */
void lko_score_dense(const float *q, int64_t n_items, int k, const float *u, float *out)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n_items; i++) {
        const float *qi = q + i * k;
        float acc = 0.0f;
        for (int f = 0; f < k; f++) acc = fmaf(qi[f], u[f], acc);
        out[i] = acc;
    }
}

/* ------------------------------------------------------------------------- */
/* Batches of queries (bench.py's cpu_baseline / at-scale parity legs)          */
/* ------------------------------------------------------------------------- */

/* n_q recommend queries, each the reference's per-query path: scores = Q @ u
 * (ALSBase.__call__, src/lenskit/als/_common.py:163-170; here the fixed k-ordered chain of
 * lko_score_dense), the query's own items struck out (candidates = all items minus the
 * user's, src/lenskit/basic/candidates.py:77-94), heap top-n (lko_argtopn).  Queries are
 * independent and run on OpenMP threads, as the reference's batch runner spreads them over
 * worker processes (src/lenskit/batch/_runner.py:259-345).  out_idx [n_q x n] is padded with
 * -1, out_scores with NaN. */
int lko_score_topn_batch(const float *q, int64_t n_items, int k, const float *users, int64_t n_q,
                         const int64_t *ex_ptr, const int32_t *ex_idx, int n, int32_t *out_idx,
                         float *out_scores, int n_threads)
{
#ifdef _OPENMP
    if (n_threads <= 0 || n_threads > LKO_MAX_THREADS) n_threads = lko_num_threads();
#else
    n_threads = 1;
#endif
#pragma omp parallel num_threads(n_threads)
    {
        float *sc = (float *)malloc(sizeof(float) * (size_t)n_items);
#pragma omp for schedule(dynamic, 4)
        for (int64_t b = 0; b < n_q; b++) {
            const float *u = users + b * k;
            for (int64_t i = 0; i < n_items; i++) {
                const float *qi = q + i * k;
                float acc = 0.0f;
                for (int f = 0; f < k; f++) acc = fmaf(qi[f], u[f], acc);
                sc[i] = acc;
            }
            if (ex_ptr)
                for (int64_t e = ex_ptr[b]; e < ex_ptr[b + 1]; e++)
                    if (ex_idx[e] >= 0 && ex_idx[e] < n_items) sc[ex_idx[e]] = NAN;
            int32_t *oi = out_idx + b * n;
            int64_t m = lko_argtopn(sc, NULL, n_items, n, oi);
            for (int64_t j = 0; j < n; j++) {
                out_scores[b * n + j] = j < m ? sc[oi[j]] : NAN;
                if (j >= m) oi[j] = -1;
            }
        }
        free(sc);
    }
    return 0;
}

/* n_q item-kNN score queries (lko_iknn_score each), CSR-style history / target lists. */
int lko_iknn_score_batch(const int64_t *s_ptr, const int32_t *s_idx, const float *s_val,
                         int64_t n_items, int64_t n_q, const int64_t *ref_ptr,
                         const int32_t *ref_items, const float *ref_rates, const int64_t *tgt_ptr,
                         const int32_t *tgt_items, int max_nbrs, int min_nbrs, int explicit_,
                         float *out_scores, uint8_t *out_valid, int32_t *out_counts,
                         int n_threads)
{
    int bad = 0;
#ifdef _OPENMP
    if (n_threads <= 0 || n_threads > LKO_MAX_THREADS) n_threads = lko_num_threads();
#else
    n_threads = 1;
#endif
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads) reduction(| : bad)
    for (int64_t b = 0; b < n_q; b++) {
        bad |= lko_iknn_score(s_ptr, s_idx, s_val, n_items, ref_items + ref_ptr[b],
                              explicit_ ? ref_rates + ref_ptr[b] : ref_rates,
                              ref_ptr[b + 1] - ref_ptr[b], tgt_items + tgt_ptr[b],
                              tgt_ptr[b + 1] - tgt_ptr[b], max_nbrs, min_nbrs, explicit_,
                              out_scores + tgt_ptr[b], out_valid + tgt_ptr[b],
                              out_counts + tgt_ptr[b]);
    }
    return bad;
}
