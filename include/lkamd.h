/*
 * lkamd.h -- C ABI of the MI355X (gfx950) backend for LensKit's hot path.
 *
 * Every entry point replaces one function of the reference's native module
 * `lenskit._accel` (Rust/PyO3) or the NumPy call next to it; the reference
 * interface each one stands in for is cited as file:line relative to the
 * lenskit/lkpy checkout.  Plain pointers and sizes only: no torch / Arrow /
 * Python types cross this boundary.  INTEGRATION.md shows the binding a
 * LensKit maintainer would add on the reference side.
 *
 * Conventions
 *  - `d_` pointers are DEVICE pointers (HBM); `h_` pointers are HOST pointers.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  All
 *    device entry points are asynchronous on that stream unless stated.
 *  - Return value: 0 on success, negative LK_E_* code on failure;
 *    lk_last_error() returns a thread-local message for the last failure.
 *  - CSR: `indptr` has n_rows+1 entries, int32 or int64 (`indptr_is_64`);
 *    `indices` int32; `values` float32.  This is the layout of the
 *    reference's SparseRowArray (src/lenskit/data/matrix.py:318-539; Rust view
 *    src/accel/sparse/csr.rs:44-223): Arrow List<Struct{index:i32,value:f32}>
 *    (int32 offsets) or LargeList (int64 offsets).
 *  - Factor matrices are row-major float32 with an explicit leading dimension
 *    `ld` (floats).  The kernels require ld == lk_padded_dim(k) and the pad
 *    columns [k, ld) to be zero; lk_pad_rows / lk_unpad_rows convert.
 */
#ifndef LKAMD_H
#define LKAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LK_OK 0
#define LK_E_INVALID (-1)   /* bad argument (shape, k, null pointer) */
#define LK_E_HIP (-2)       /* HIP runtime error */
#define LK_E_NOT_SPD (-3)   /* ALS normal matrix not positive definite
                               (reference: RuntimeError("ALS solve error: ..."),
                               src/accel/als/implicit.rs:79, solve.rs:99-105) */
#define LK_E_NAN_SIM (-4)   /* NaN similarity (reference: ValueError("similarity is
                               null"), src/accel/knn/accum.rs:146-151) */
#define LK_E_NOMEM (-5)
#define LK_E_CANCELLED (-6) /* cooperative cancel (AccelTask.cancel(),
                               src/accel/tasks/mod.rs:88-95) */

#define LK_SOLVER_CHOLESKY 0 /* exact SPD solve: the reference's method (LAPACK sposv) */
#define LK_SOLVER_CG 1       /* tolerance-terminated conjugate gradient */
#define LK_SOLVER_AUTO 2     /* = Cholesky: exact at every k <= 256, like the reference */

const char *lk_last_error(void);
const char *lk_version(void);
/* Number of visible HIP devices (0 when there is no GPU / no driver). */
int lk_device_count(void);

/* Padded embedding width used on the device for `k` features: the next of
 * {16, 32, 64, 128, 256}; for 256 < k <= 1024 the next multiple of 64 (those sizes are solved on
 * tiles kept in HBM, csrc/als_big.hip: the reference's `POSV::solve`,
 * src/accel/als/solve.rs:65-107, takes any k); 0 if k is unsupported (k < 1 or k > 1024). */
int32_t lk_padded_dim(int32_t k);

/* Copy an [n x k] row-major matrix (leading dimension ld_src) into an
 * [n x ld_dst] one, zero-filling columns [k, ld_dst); and the inverse. */
int lk_pad_rows(const float *d_src, int64_t n, int32_t k, int32_t ld_src, float *d_dst,
                int32_t ld_dst, void *stream);
int lk_unpad_rows(const float *d_src, int64_t n, int32_t k, int32_t ld_src, float *d_dst,
                  int32_t ld_dst, void *stream);

/* ------------------------------------------------------------------------
 * Task control: cooperative cancel + live progress for the long-running entry points.
 * Replaces the `AccelTask` pyclass protocol (src/accel/tasks/mod.rs:33-106: `cancel()` sets a
 * flag the rayon workers poll, `current_progress()` reads an atomic row counter) that
 * `run_accel_task` drives from the main thread while `invoke` runs on a helper thread
 * (src/lenskit/parallel/_task.py:34-57).
 *   - the control block owns two words in pinned, device-mapped host memory: `cancel`
 *     (lk_task_ctl_cancel stores 1; callable from any thread while kernels run) and
 *     `rows_done` (the kernels post their running count there every 256 units;
 *     lk_task_ctl_progress reads it without touching the device or the stream);
 *   - attach it with lk_als_plan_set_ctl / lk_iknn_plan_set_ctl (NULL detaches); attached
 *     kernels skip every row (task) that has not started once the cancel is seen, and the
 *     entry point that synchronises next -- lk_als_check_status, lk_iknn_build_count,
 *     lk_iknn_build_fill -- returns LK_E_CANCELLED (outputs are then unspecified, like a
 *     cancelled rayon job's);
 *   - progress unit = rows (tasks/mod.rs:97-105); exact after the synchronising call.
 * Without a control block the kernels carry no polling code at all.
 * ---------------------------------------------------------------------- */
typedef struct lk_task_ctl lk_task_ctl;
int lk_task_ctl_create(lk_task_ctl **out);
void lk_task_ctl_destroy(lk_task_ctl *ctl);
void lk_task_ctl_cancel(lk_task_ctl *ctl);
int lk_task_ctl_cancelled(const lk_task_ctl *ctl);
/* clear the cancel flag and the progress count (before reusing the block) */
void lk_task_ctl_reset(lk_task_ctl *ctl);
int lk_task_ctl_progress(const lk_task_ctl *ctl, int64_t *rows_done, int64_t *rows_total);

/* ------------------------------------------------------------------------
 * Gramian:  out = M^T M + reg * I          (k x k, row-major, ld_out floats)
 * Replaces `_implicit_otor` (src/lenskit/als/_implicit.py:177-184), a NumPy
 * sgemm in the reference.  `d_ws` must hold lk_gramian_workspace_bytes(k).
 * Deterministic (fixed reduction order); the result is exactly symmetric.
 * ---------------------------------------------------------------------- */
size_t lk_gramian_workspace_bytes(int32_t k);
int lk_gramian(const float *d_m, int64_t n, int32_t k, int32_t ld, float reg, float *d_out,
               int32_t ld_out, void *d_ws, void *stream);

/* ------------------------------------------------------------------------
 * Implicit-feedback ALS half-epoch.
 * Replaces `lenskit._accel.als.train_implicit_matrix(matrix, this, other, otor)`
 * (src/lenskit/_accel/als.pyi:11-16; src/accel/als/implicit.rs:35-125 with
 * src/accel/als/solve.rs:65-107):  for every CSR row r
 *     A = otor + sum_j v_j q_j q_j^T,  y = sum_j (v_j + 1) q_j,  x = A^-1 y,
 *     this[r] <- x        (empty row: this[r] <- 0, contributes 0 to the delta)
 * and *d_out_frob = sqrt(sum_r ||x - this_old[r]||^2)  (float32, on device).
 *
 * A plan (row schedule: longest-first order, split of very long rows into
 * chunks, workspace layout) is built ONCE per matrix from the HOST copy of
 * indptr and reused for every epoch.
 * ---------------------------------------------------------------------- */
typedef struct lk_als_plan lk_als_plan;

/* Default plan = LK_ALS_PLAN_HYBRID_ORDER (round 5), unless the environment says
 * LK_ALS_RHS_ORDER=accurate (flags 0) or =reference (LK_ALS_PLAN_REFERENCE_ORDER). */
int lk_als_plan_create(lk_als_plan **out, const void *h_indptr, int indptr_is_64, int64_t n_rows,
                       int32_t k, int32_t solver);
/* The same with explicit flags (0 = the tuned kernels' own "accurate" summation everywhere).
 *
 * The reference's ARITHMETIC ORDER on long rows: `mtl.dot(&o_picked)`
 * (src/accel/als/implicit.rs:112) is matrixmultiply's sgemm, which sums the row's entries in
 * blocks of KC = 256 -- one fma chain per block, the block sums added one after the other -- and
 * `mt.dot(&vals)` (implicit.rs:117) is ONE sequential float32 chain per feature.  On rows of
 * 10^4 .. 10^6 entries those two sums drift systematically (1e-4 .. 7e-2 from the float64 sums),
 * so a kernel that sums more accurately lands that far from the REFERENCE's factors.
 *
 * LK_ALS_PLAN_HYBRID_ORDER (default): rows with more than LK_ALS_REF_LEN entries (environment,
 * default 2048 -- the rows that are pre-reduced in chunks anyway) are evaluated in the reference's
 * order: 256-entry chunks (one MFMA fmaf chain each, from zero), the chunk slabs added one after
 * the other in chunk order, OtOr last, y by the sequential chain of csrc/als_rhs.hip (bit-identical
 * to the reference's y).  Shorter rows keep the tuned order (the two agree to ~1e-5 there).  The
 * y buffer is part of the plan's workspace; works with task control, CSR views with row offsets
 * and any row / column relabelling (the chain runs over the row's entries as the CSR lists them:
 * give the rows in the reference's entry order -- ascending column of the ORIGINAL labelling).
 * k <= 256, exact solver; other plans ignore the flag.
 *
 * LK_ALS_PLAN_REFERENCE_ORDER: the strict variant -- EVERY row of more than 256 entries in
 * 256-entry chunks and every dense row's y by the chain; takes y from
 * lk_als_plan_set_rhs_workspace ([n_rows x lk_padded_dim(k)], mandatory for it), no task control.
 * tests/test_gpu_als_rhs_order.py. */
#define LK_ALS_PLAN_REFERENCE_ORDER 1
#define LK_ALS_PLAN_HYBRID_ORDER 2
int lk_als_plan_create_ex(lk_als_plan **out, const void *h_indptr, int indptr_is_64,
                          int64_t n_rows, int32_t k, int32_t solver, int32_t flags);
void lk_als_plan_destroy(lk_als_plan *plan);
/* Rows of the plan that are pre-reduced in chunks (longer than 2048 entries; LK_ALS_REF_LEN for
 * hybrid plans; 256 for strict reference-order plans): the first tasks of the longest-first
 * order, i.e. the rows sorted by descending length, ties in row order. */
int64_t lk_als_plan_long_rows(const lk_als_plan *plan);
/* Hybrid plans: the right-hand sides of those rows as the last half-epoch run with workspace
 * `d_ws` formed them in the reference's order -- [lk_als_plan_long_rows x lk_padded_dim(k)]
 * floats, row t = the t-th longest row (device pointer into d_ws; NULL for other plans).
 * Diagnostic / test access: tests/test_gpu_als_rhs_order.py holds it to the chain bit for bit. */
const float *lk_als_plan_yref(const lk_als_plan *plan, const void *d_ws);
/* Device workspace the half-epoch needs (bytes); allocate once, reuse. */
size_t lk_als_plan_workspace_bytes(const lk_als_plan *plan);
/* Effective solver of the plan (LK_SOLVER_CHOLESKY or LK_SOLVER_CG). */
int32_t lk_als_plan_solver(const lk_als_plan *plan);

/* CG controls (ignored by the Cholesky solver): stop when ||r|| <= tol*||y|| or
 * after max_iter iterations (<=0: k iterations).  Defaults 1e-7 / k. */
int lk_als_plan_set_cg(lk_als_plan *plan, float tol, int32_t max_iter);
/* CG iterations spent and non-empty rows solved by the LAST CG half-epoch run with workspace
 * d_ws (sum over rows; blocking).  Diagnostic: the roofline of the CG kernel is
 * iterations * (4k flops per entry of the row + 2k^2). */
int lk_als_plan_cg_stats(const lk_als_plan *plan, void *d_ws, void *stream,
                         int64_t *out_iterations, int64_t *out_rows);
/* Attach a task-control block (cancel / progress) to every half-epoch run with this plan. */
int lk_als_plan_set_ctl(lk_als_plan *plan, lk_task_ctl *ctl);
/* Short rows at large k (implicit model, padded k = 128 / 256).  A row with n <= 64 entries
 * is a rank-n update of OtOr, the same matrix for every row of the half-epoch: with
 * Z = other * OtOr^-1 ([n_cols x lk_padded_dim(k)], pad columns zero) supplied by the caller,
 * lk_als_implicit_half_epoch solves those rows through the Woodbury identity (an n x n
 * system, O(n^2 k) instead of the k^3/3 of `sposv`, src/accel/als/solve.rs:65-107) -- the
 * same solution in exact arithmetic, no iteration (n <= 4 / 8: four / two rows per wave in one
 * 16 x 16 tile; n <= 16: one 16 x 16 system per wave; 17 .. 64: a 32 x 32 or 64 x 64 system on the
 * k <= 64 solver; 65 .. 128 at padded k = 256: a 128 x 128 system on the k = 128 blocked solver).  lk_als_plan_short_rows: how many rows of
 * the plan have <= 16 entries.  lk_als_plan_set_z: Z for the NEXT half-epoch calls (NULL: every row
 * takes the dense solve); the buffer must stay alive until those calls have finished. */
int64_t lk_als_plan_short_rows(const lk_als_plan *plan);
/* rows the 16 x 16 ... 64 x 64 Woodbury systems take: <= 64 entries at padded k = 256 (rows of
 * 65 .. 128 entries additionally take a 128 x 128 system there unless LK_ALS_WB128=0); at padded
 * k = 128 <= 64 / 32 / 16 entries with LK_ALS_WB64_K128 = 64 (default) / 32 / 0 */
int64_t lk_als_plan_woodbury_rows(const lk_als_plan *plan);
int lk_als_plan_set_z(lk_als_plan *plan, const float *d_z);
/* The same path with NOTHING computed on the caller's side: hand the plan a device buffer of
 * n_cols x lk_padded_dim(k) floats and every lk_als_implicit_half_epoch run with it forms Z
 * itself on the launch stream -- OtOr^-1 to float64 accuracy by an in-register sweep + two
 * Newton-Schulz steps (csrc/spd_inverse.hip), then Z = other * OtOr^-1 on the scoring GEMM.  If
 * OtOr turns out not to be positive definite (reg = 0 with rank-deficient factors) the decision
 * is taken ON THE DEVICE: the Woodbury kernels stand down and a dense fallback launch solves
 * their rows exactly as `sposv` would (and reports a row whose own matrix is not positive
 * definite the same way).  No library call, no host synchronisation.  NULL detaches the buffer. */
int lk_als_plan_set_z_workspace(lk_als_plan *plan, float *d_zbuf);
/* Strict reproduction of the reference's right-hand side.  `train_row_solve` forms
 * y = mt.dot(&vals) on a TRANSPOSED (strided) view (src/accel/als/implicit.rs:116-117;
 * explicit model: src/accel/als/explicit.rs:110), which ndarray evaluates as ONE sequential
 * float32 chain per feature over the row's entries, product and sum rounded separately.  On rows
 * of 10^5 .. 10^6 entries that chain drifts 1e-4 .. 7e-2 from the exact sum; the solve kernels'
 * own (slotted / chunked) sum does not, so on such rows the default result is CLOSER TO FLOAT64
 * THAN THE REFERENCE IS and can be > 1e-4 away from it.  Hand the plan a device buffer of
 * n_rows x lk_padded_dim(k) floats and every half-epoch (implicit and explicit) first forms y in
 * exactly the reference's order (csrc/als_rhs.hip: lane = feature, entries in row order) and the
 * dense solve kernels take their right-hand side from it -- bit-identical y, so those rows land
 * within 1e-4 of the reference.  Slower (one latency-bound chain per feature); rows served by
 * the Woodbury kernels (<= 64 entries at padded k > 64) never form y and are unaffected; ignored
 * while a task-control block is attached.  NULL detaches (default: the accurate sum). */
int lk_als_plan_set_rhs_workspace(lk_als_plan *plan, float *d_y);
/* Several plans over ROW SLICES of one half-epoch (the sharded engine cuts a rank's rows into
 * slices so that the all-gather of one slice runs under the solve of the next) share one Z: the
 * leading slice's plan owns the buffer (lk_als_plan_set_z_workspace) and is launched first; the
 * others are given that buffer and the device address of the leader's "OtOr is not positive
 * definite" flag (lk_als_plan_z_flag(leader, leader's workspace)), which they copy into their
 * own status word at every launch -- same stream, so it is final by then.  NULL / NULL detaches. */
int lk_als_plan_set_z_shared(lk_als_plan *plan, const float *d_z, const void *d_flag);
int lk_als_plan_set_z_leader(lk_als_plan *plan, int on); /* form Z even without short rows of its own */
const void *lk_als_plan_z_flag(const lk_als_plan *plan, const void *d_ws);
/* OtOr^-1 alone (diagnostics / tests): d_out [KP x KP] floats zero padded, *d_flag = 0 or != 0
 * when d_a is not positive definite, d_ws lk_spd_inverse_workspace_bytes(k) bytes; padded
 * k = 128 / 256 only. */
size_t lk_spd_inverse_workspace_bytes(int32_t k);
int lk_spd_inverse(const float *d_a, int32_t lda, int32_t k, float *d_out, int32_t *d_flag,
                   void *d_ws, void *stream);

int lk_als_implicit_half_epoch(const lk_als_plan *plan, const void *d_indptr,
                               const int32_t *d_indices, const float *d_values, int64_t n_rows,
                               int64_t n_cols, int32_t k, float *d_this, int32_t ld_this,
                               const float *d_other, int32_t ld_other, const float *d_otor,
                               int32_t ld_otor, void *d_ws, float *d_out_frob, void *stream);
/* Explicit-feedback (biased-MF) half-epoch: replaces `_accel.als.train_explicit_matrix(matrix,
 * this, other, reg)` (src/lenskit/_accel/als.pyi, src/accel/als/explicit.rs:33-119):
 *   A = sum_j q_j q_j^T + reg * n * I,  A x = sum_j r_j q_j   (r = bias-normalised ratings),
 * empty rows -> zeros; same plan, workspace, in-place update, status word and return value
 * (sqrt of the summed squared row deltas at d_out_frob) as the implicit form.  Exact solver
 * only. */
int lk_als_explicit_half_epoch(const lk_als_plan *plan, const void *d_indptr,
                               const int32_t *d_indices, const float *d_values, int64_t n_rows,
                               int64_t n_cols, int32_t k, float *d_this, int32_t ld_this,
                               const float *d_other, int32_t ld_other, float reg, void *d_ws,
                               float *d_out_frob, void *stream);
/* Optional per-kernel timing with HIP events recorded on the launch stream
 * (bench.py's roofline leg).  get_timing waits for the recorded events, returns the
 * summed durations (ms) of the chunk kernel and of the solve kernel over the
 * *n_launches half-epochs recorded since the last call (at most 128), and resets. */
int lk_als_plan_enable_timing(lk_als_plan *plan, int enable);
int lk_als_plan_get_timing(lk_als_plan *plan, double *ms_chunk, double *ms_solve,
                           int32_t *n_launches);

/* Synchronise `stream` and translate the device status word of the last
 * half-epoch into a return code (LK_E_NOT_SPD with the offending row in
 * lk_last_error()).  Call before trusting `this`. */
int lk_als_check_status(const lk_als_plan *plan, void *d_ws, void *stream);

/* Host-pointer convenience form with the reference's exact argument list
 * (`train_implicit_matrix(matrix, this, other, otor)`, src/accel/als/implicit.rs:35-84:
 * this: [n_rows x k] updated in place, other: [n_cols x k], otor: [k x k], all C-contiguous
 * host float32; offsets int32 or int64).  Allocates, copies, runs (default plan: hybrid
 * summation order; padded k = 128 / 256: the Woodbury kernels for the short rows when there
 * are >= LK_ALS_WB_MIN_ROWS of them), copies back; blocking.  Errors: LK_E_NOT_SPD with
 * "ALS solve error: ..." in lk_last_error() (implicit.rs:79), `this` left untouched.
 * tests/test_gpu_host_abi.py runs INTEGRATION.md's binding sketch through it with raw ctypes. */
int lk_als_implicit_half_epoch_host(const void *h_indptr, int indptr_is_64,
                                    const int32_t *h_indices, const float *h_values,
                                    int64_t n_rows, int64_t n_cols, int32_t k, float *h_this,
                                    const float *h_other, const float *h_otor, int32_t solver,
                                    float *h_out_frob);
/* The same with the reference's task controls (src/accel/tasks/mod.rs:62-106,
 * src/lenskit/parallel/_task.py:25-57): `ctl` (may be NULL) is polled by the running kernels --
 * lk_task_ctl_cancel from another thread makes the call return LK_E_CANCELLED with the rows
 * solved so far written to `this` (the reference updates `this` in place row by row as well);
 * lk_task_ctl_progress reads the live count of finished rows.  With a control block every row
 * takes the dense kernels (the Woodbury kernels do not poll). */
int lk_als_implicit_half_epoch_host_ctl(const void *h_indptr, int indptr_is_64,
                                        const int32_t *h_indices, const float *h_values,
                                        int64_t n_rows, int64_t n_cols, int32_t k, float *h_this,
                                        const float *h_other, const float *h_otor,
                                        int32_t solver, float *h_out_frob, lk_task_ctl *ctl);

/* ------------------------------------------------------------------------
 * Item-item similarity build.
 * Replaces `lenskit._accel.knn.compute_similarities(ui, iu, shape, min_sim,
 * save_nbrs)` (src/lenskit/_accel/knn.pyi:8-14; src/accel/knn/item_train.rs:33-152
 * + src/accel/sparse/consumer.rs:24-142): for each item i,
 *     dots[j] = sum over users u of i (ascending u) of a_ui * a_uj   (j != i)
 * accumulated in f32 as round(a_ui*a_uj) then add -- the reference's order and
 * rounding, so values are bit-identical -- keep dots[j] >= min_sim, optional
 * per-row top-`save_nbrs` (by similarity, ties by first encounter), rows sorted
 * by column.  Output CSR has int64 offsets (LargeList).
 *
 * Two-phase because the output size is data dependent:
 *   lk_iknn_build_count  -> fills d_out_indptr (n_items+1, int64, exclusive scan)
 *                           and returns the total through *h_total_nnz (blocking);
 *   lk_iknn_build_fill   -> writes d_out_indices / d_out_values.
 * Both calls take the SAME workspace (it carries the per-task counts / offsets
 * between them).  When n_items^2 (index, value) pairs fit comfortably in free HBM
 * (cap: env LK_IKNN_STAGE_GB, default 64) the plan reserves a staging area in the
 * workspace: the count call then does the whole computation once and the fill call
 * only compacts; otherwise each call is a full pass over the data.
 * ---------------------------------------------------------------------- */
typedef struct lk_iknn_plan lk_iknn_plan;
int lk_iknn_plan_create(lk_iknn_plan **out, const void *h_ui_indptr, const void *h_iu_indptr,
                        int indptr_is_64, int64_t n_users, int64_t n_items);
/* Shard of the build: output rows (items) [row_begin, row_end) only -- rows are independent
 * (`compute_similarities` is a par_iter over rows, item_train.rs:56-70), so ranks of a
 * multi-GPU job each build a row block with no collective.  d_out_indptr then has
 * (row_end - row_begin + 1) entries; column numbers stay global. */
int lk_iknn_plan_create_rows(lk_iknn_plan **out, const void *h_ui_indptr,
                             const void *h_iu_indptr, int indptr_is_64, int64_t n_users,
                             int64_t n_items, int64_t row_begin, int64_t row_end);
void lk_iknn_plan_destroy(lk_iknn_plan *plan);
int lk_iknn_plan_set_ctl(lk_iknn_plan *plan, lk_task_ctl *ctl);
/* Optional timing of the similarity kernel with HIP events recorded on the launch stream
 * (bench.py's roofline leg): get_timing waits for the events, returns the summed duration
 * (ms) of the build-kernel launches recorded since the last call (one for a staged build,
 * two for a two-pass build) and resets. */
int lk_iknn_plan_enable_timing(lk_iknn_plan *plan, int enable);
int lk_iknn_plan_get_timing(lk_iknn_plan *plan, double *ms_build, int32_t *n_launches);
size_t lk_iknn_plan_workspace_bytes(const lk_iknn_plan *plan);

int lk_iknn_build_count(const lk_iknn_plan *plan, const void *d_ui_indptr,
                        const int32_t *d_ui_indices, const float *d_ui_values,
                        const void *d_iu_indptr, const int32_t *d_iu_indices,
                        const float *d_iu_values, float min_sim, int64_t save_nbrs, void *d_ws,
                        int64_t *d_out_indptr, int64_t *h_total_nnz, void *stream);
int lk_iknn_build_fill(const lk_iknn_plan *plan, const void *d_ui_indptr,
                       const int32_t *d_ui_indices, const float *d_ui_values,
                       const void *d_iu_indptr, const int32_t *d_iu_indices,
                       const float *d_iu_values, float min_sim, int64_t save_nbrs, void *d_ws,
                       const int64_t *d_out_indptr, int32_t *d_out_indices, float *d_out_values,
                       void *stream);

/* ------------------------------------------------------------------------
 * Dense scoring with fused top-K.
 * Replaces, for a batch of users, `ALSBase.__call__` scoring
 * (src/lenskit/als/_common.py:159-170: scores = Q u) followed by
 * `TopNRanker` -> `ItemList.top_n` -> `_accel.data.argtopn`
 * (src/lenskit/basic/topn.py:45-69, src/lenskit/data/_items.py:942-998,
 * src/accel/data/sorting.rs:132-172).
 *   scores[b][i] = fma-chain over f = 0..k-1 of U[b][f]*Q[i][f]  (f32 MFMA: exact,
 *   k-ordered) for every item i not listed in the exclusion CSR row b
 *   (candidate selector: training items minus the query's items,
 *   src/lenskit/basic/candidates.py:77-94); NaN scores are skipped;
 *   out_idx[b][0..n) = indices of the n largest scores, descending, ties by
 *   lower index; out_score likewise; rows with fewer than n candidates are
 *   padded with index -1 / score NaN.
 * Two implementations, identical results bit for bit:
 *   - fused (n <= 128, >= 16384 items, >= 8192 users, k > 16; env LK_TOPK_FUSED_MIN_ITEMS /
 *     LK_TOPK_FUSED_MIN_USERS): a threshold per user from an item sample, then ONE pass of
 *     the contraction whose epilogue keeps only the entries that reach it, then the exact order
 *     among those candidates -- the n_users x n_items score matrix never exists in memory;
 *     this path synchronises the stream before returning (it reads the list of rows that did
 *     not end up with n valid candidates -- overflowing lists, thresholds that proved too
 *     high -- and redoes exactly those rows through the panel path);
 *   - panel: 2048 users at a time are scored into the workspace and selected from there.
 * Workspace (ask lk_score_topk_workspace_bytes, never assume): the fused path works on batches
 * of up to 262 144 users (LK_TOPK_FUSED_ROWS; 65 536 before round 3) and holds per batch
 * 16 KiB of candidate lists per user (<= 4 GiB) plus the stage-1 sample panel,
 * users x ceil(n_items / 16) floats, which is what bounds the batch: never more than 16 GiB
 * (for catalogues so large that 8192 sample rows would exceed it the batch shrinks below 8192
 * rather than the panel growing).  Ceiling: ~20.5 GiB, reached only with >= 262 144 users and
 * >= 250 000 items; ML-25M (162 541 x 62 423): 5.3 GiB.  The panel path needs 2048 x n_items
 * floats.
 * ---------------------------------------------------------------------- */
size_t lk_score_topk_workspace_bytes(int64_t n_users, int64_t n_items, int32_t n);
int lk_score_topk(const float *d_users, int32_t ld_users, int64_t n_users, const float *d_items,
                  int32_t ld_items, int64_t n_items, int32_t k, int32_t n,
                  const int64_t *d_excl_ptr, const int32_t *d_excl_items, void *d_ws,
                  int32_t *d_out_idx, float *d_out_score, void *stream);

/* Scores only: out[b][i] = the same exact k-ordered chain, for every (user, item) pair
 * (`ALSBase.__call__`, src/lenskit/als/_common.py:159-170). */
int lk_score_dense(const float *d_users, int32_t ld_users, int64_t n_users, const float *d_items,
                   int32_t ld_items, int64_t n_items, int32_t k, float *d_out, int64_t ld_out,
                   void *stream);

/* Plain top-N over score vectors already in HBM (one row per query), the direct stand-in for
 * `_accel.data.argtopn(scores, n)` (src/accel/data/sorting.rs:132-172) and, with n < 0, for
 * `_accel.data.argsort_descending(scores)` (sorting.rs:69-103): every valid entry by descending
 * score.  d_out_idx is [n_rows x min(n, row_len)] (n < 0: [n_rows x row_len]), -1 padded.
 * n <= 4096 runs the selection kernel and needs no workspace; n < 0 or n > 4096 is a full
 * stable sort and needs lk_argtopn_workspace_bytes(n_rows, row_len, n) bytes at d_ws.
 * (lk_score_topk likewise: n < 0 ranks all candidates, output [n_users x n_items].) */
size_t lk_argtopn_workspace_bytes(int64_t n_rows, int64_t row_len, int32_t n);
int lk_argtopn(const float *d_scores, int64_t n_rows, int64_t row_len, int32_t n, void *d_ws,
               int32_t *d_out_idx, void *stream);

/* Per-row top-`save_nbrs` truncation of a built similarity matrix -- the second half of
 * `sim_row` (src/accel/knn/item_train.rs:139-151): keep the save_nbrs most similar
 * neighbours of every row, ties at the cut in order of first encounter (first shared user,
 * then column), rows stay sorted by column.  lk_iknn_build_* take save_nbrs <= 0 (none);
 * run these on their output.  `nnz` = entries of the input matrix.  _count is blocking. */
size_t lk_iknn_truncate_workspace_bytes(int64_t n_rows, int64_t nnz);
/* n_rows rows of the similarity matrix; row r is item (row_begin + r) (0 for a full build) */
int lk_iknn_truncate_count(const int64_t *d_sim_indptr, const int32_t *d_sim_indices,
                           const float *d_sim_values, const void *d_iu_indptr,
                           int iu_indptr_is_64, const int32_t *d_iu_indices, int64_t n_rows,
                           int64_t row_begin, int64_t nnz, int64_t save_nbrs, void *d_ws,
                           int64_t *d_out_indptr, int64_t *h_total_nnz, void *stream);
int lk_iknn_truncate_fill(const int64_t *d_sim_indptr, const int32_t *d_sim_indices,
                          const float *d_sim_values, int64_t n_rows, int64_t nnz, void *d_ws,
                          const int64_t *d_out_indptr, int32_t *d_out_indices,
                          float *d_out_values, void *stream);

/* ------------------------------------------------------------------------
 * Item-kNN "recommend" for a BATCH of queries: score EVERY item and keep the top n.
 * Replaces, for a batch, what `pipelines/iknn-explicit.toml`'s recommender runs per query
 * (src/lenskit/batch/_runner.py:283-308): the candidate selector (all training items minus the
 * query's own, src/lenskit/basic/candidates.py:77-94), `ItemKNNScorer.__call__` over the
 * candidates (src/lenskit/knn/item.py:231-295 -> `score_explicit` / `score_implicit`,
 * src/accel/knn/item_score.rs:23-111, accum.rs), the item means added back (item.py:282), and
 * `TopNRanker` (src/lenskit/basic/topn.py:45-69 -> src/accel/data/sorting.rs:132-172).
 *   query q: reference items ref_items[ref_ptr[q]..ref_ptr[q+1]) (negative = null, skipped) with
 *   centred ratings ref_rates (NULL => implicit feedback); d_item_bias NULL or [n_items] floats
 *   added to every score (one f32 add, as NumPy's `scores + means`); exclude_refs != 0 strikes
 *   the query's own items from the candidates; items with fewer than min_nbrs contributors have
 *   no score and are never listed.  out_idx [n_queries x n] = the n best scored items,
 *   descending, ties by lower item number, -1 padded; out_score likewise (NaN padded; may be
 *   NULL).  n < 0: every scored candidate, ranked (output [n_queries x n_items]).
 *   The SCORES are the reference accumulator's bit for bit (csrc/iknn_recommend.hip: one wave
 *   per (query, window of 4096 items); round 6: every weight is added to its item's LDS cell by
 *   ds_add_f32 in history order -- the LDS applies same-address adds in lane order, probed on the
 *   device --, items with more than max_nbrs contributors are found again and replayed on the
 *   reference's heap by a kernel of their own; the list kernel of rounds 4-5 -- hits laid out per
 *   item through in-order LDS cursors -- is the fallback.  No barrier per history row as in
 *   lk_iknn_score_batch's slot kernel).
 *   h_query_hits (HOST, [n_queries]): per query, the summed lengths of its reference items'
 *   similarity rows (the caller has the row lengths: `item_counts`); max_query_hits >= their
 *   maximum sizes the workspace (the list kernel's hit region, the queue of heap items: at most
 *   hits / (max_nbrs + 1) per query).  Queries are processed in batches of <= 6144
 *   (LK_REC_PANEL_ROWS), <= 2^28 hits and a full queue at most; a call of several batches keeps
 *   two score panels (LK_REC_PANEL_GB bounds them) and runs a batch's selection on a side stream
 *   while the next batch is scored.  Blocking (returns LK_E_NAN_SIM for a NaN similarity).
 * ---------------------------------------------------------------------- */
size_t lk_iknn_recommend_workspace_bytes(int64_t n_items, int64_t n_queries,
                                         int64_t max_query_hits, int32_t max_nbrs, int32_t n);
int lk_iknn_recommend(const int64_t *d_sim_indptr, const int32_t *d_sim_indices,
                      const float *d_sim_values, int64_t n_items, int64_t n_queries,
                      const int64_t *d_ref_ptr, const int32_t *d_ref_items,
                      const float *d_ref_rates, const float *d_item_bias, int32_t max_nbrs,
                      int32_t min_nbrs, int32_t n, int exclude_refs, const int64_t *h_query_hits,
                      int64_t max_query_hits, void *d_ws, int32_t *d_out_idx, float *d_out_score,
                      void *stream);

/* ------------------------------------------------------------------------
 * Item-kNN scoring for a BATCH of queries.
 * Replaces `_accel.knn.score_explicit` / `score_implicit`
 * (src/lenskit/_accel/knn.pyi:15-29; src/accel/knn/item_score.rs:23-111,
 * src/accel/knn/accum.rs:16-239), which score ONE query per call.
 *   query q has reference (history) items ref_items[ref_ptr[q]..ref_ptr[q+1]) with
 *   centred ratings ref_rates (NULL => implicit) and targets
 *   tgt_items[tgt_ptr[q]..tgt_ptr[q+1]); negative item numbers are nulls.
 *   score_t = sum_{top max_nbrs by sim} s*v / sum s  (explicit)  or  sum s (implicit);
 *   fewer than min_nbrs contributors => NaN.  out_counts = contributors kept
 *   (-1 for a null target).  Blocking (returns LK_E_NAN_SIM for a NaN similarity).
 *   Calls in which no query has more than 1024 targets, with max_nbrs <= 255, take the
 *   candidate-list kernel: targets hashed in LDS, all history rows streamed at once, and the
 *   reference's accumulator followed step for step (a vector, then std's BinaryHeap push / pop),
 *   so that WHICH of several equal similarities is evicted at the max_nbrs boundary and the order
 *   of the f32 sums are the reference's: the scores are bit-identical to its.  Other calls
 *   (`score every item`) take the one-history-row-at-a-time slot kernel, which runs the same
 *   accumulator on per-item slots of the workspace (bit-identical as well, slower per target).
 *   LK_KNN_SCORE_LISTS=0 forces the slot kernel.  The matrix rows must not name a column twice
 *   (LK_E_INVALID from the list kernel).  lk_knn_score_last_stats: of the calling thread's last
 *   call, {queries scored by the list kernel, by the slot kernel, longest target list} (test hook).
 * ---------------------------------------------------------------------- */
size_t lk_iknn_score_workspace_bytes(int64_t n_items, int64_t n_queries, int32_t max_nbrs);
void lk_knn_score_last_stats(int64_t *out3);
/* 2 / 1 / 0: the last lk_iknn_recommend call of this process ran the ACCUMULATING kernel (round 6,
 * the default: weights added to their target's LDS cell by ds_add_f32 -- same-address adds of an
 * instruction applied in lane order with the VALU's rounding, probed on the device at first use),
 * the list kernel walking the similarity rows PACKED (64 entries of several (row x window) pieces
 * per instruction; needs same-address LDS atomics of one instruction served in lane order, probed
 * as well; LK_REC_ACC=0 forces it), or the list kernel piece by piece (LK_REC_PACKED=0); -1: no
 * call yet.  Test hook.  Same scores in all three.  LK_REC_OVERLAP=0: the batches' tails (queued
 * targets, selection) on the caller's stream instead of a side stream. */
int lk_iknn_recommend_last_packed(void);
int lk_iknn_score_batch(const int64_t *d_sim_indptr, const int32_t *d_sim_indices,
                        const float *d_sim_values, int64_t n_items, int64_t n_queries,
                        const int64_t *d_ref_ptr, const int32_t *d_ref_items,
                        const float *d_ref_rates, const int64_t *d_tgt_ptr,
                        const int32_t *d_tgt_items, int32_t max_nbrs, int32_t min_nbrs,
                        void *d_ws, float *d_out_scores, int32_t *d_out_counts, void *stream);

/* ------------------------------------------------------------------------
 * User-kNN scoring for a BATCH of queries.
 * Replaces `_accel.knn.user_score_items_explicit / _implicit(tgt_items, nbr_rows, nbr_sims,
 * ratings, max_nbrs, min_nbrs)` (src/accel/knn/user_score.rs:21-98): query q has neighbours
 * nbr_rows[nbr_ptr[q]..nbr_ptr[q+1]) (rows of the users x items ratings CSR, int64 offsets)
 * with similarities nbr_sims; every neighbour, in the given order, offers (weight = its
 * similarity, value = its rating) to the items it rated; per target item the `max_nbrs`
 * largest weights are kept (same ScoreAccumulator as item-kNN, accum.rs:16-239) and
 * score = sum(w * v) / sum(w)  (explicit)  or  sum(w)  (d_rat_values NULL: implicit);
 * fewer than min_nbrs contributors => NaN.  Negative neighbour rows / target items are nulls.
 * Workspace: lk_iknn_score_workspace_bytes(n_items, n_queries, max_nbrs).  Blocking.
 * ---------------------------------------------------------------------- */
int lk_uknn_score_batch(const int64_t *d_rat_indptr, const int32_t *d_rat_indices,
                        const float *d_rat_values, int64_t n_users, int64_t n_items,
                        int64_t n_queries, const int64_t *d_nbr_ptr, const int32_t *d_nbr_rows,
                        const float *d_nbr_sims, const int64_t *d_tgt_ptr,
                        const int32_t *d_tgt_items, int32_t max_nbrs, int32_t min_nbrs, void *d_ws,
                        float *d_out_scores, int32_t *d_out_counts, void *stream);

/* Neighbour similarities of user-kNN, `nbr_sims = user_vectors @ ratings`
 * (src/lenskit/knn/user.py:196) for a batch of dense query vectors:
 *   out[q][r] = sum over the entries (c, v) of CSR row r of  v * x[c][q]
 * x is [n_cols x ld_x] (column-major over queries: the queries' values of one item are
 * contiguous), out is [n_queries x ld_out].  Products accumulate in entry order (fmaf). */
int lk_csr_rows_dot(const void *d_indptr, int indptr_is_64, const int32_t *d_indices,
                    const float *d_values, int64_t n_rows, const float *d_x, int64_t ld_x,
                    int64_t n_queries, float *d_out, int64_t ld_out, void *stream);

/* Device -> pageable host memory at PCIe speed: chunks through a ring of pinned staging slots,
 * a team of `n_threads` host threads (<= 14; 0 = 8) copies the landed chunks to `h_dst` in
 * parallel (first-touching its pages on many cores).  Used for results the reference hands
 * back as host arrays -- e.g. the similarity matrix of `compute_similarities`
 * (src/accel/knn/item_train.rs:86-91): SURVEY.md section 8d counts the kNN build "to CSR sim
 * matrix on host".  Blocking; waits for `stream` (the producer of `d_src`) first. */
int lk_download(void *h_dst, const void *d_src, size_t bytes, int32_t n_threads, void *stream);
/* The same for int32 values known to lie in [0, 65536) -- the column indices of a matrix with at
 * most 65 536 columns (ML-25M: 62 423 items): they cross PCIe as uint16 (narrowed on the device
 * into d_tmp_u16, n * 2 bytes) and the host team widens them back while it copies them into
 * h_dst, so the index half of a similarity matrix costs half the link time. */
/* Pin the staging ring of lk_download now (idempotent; otherwise the first large download pays). */
int lk_download_warmup(void);
int lk_download_i32_narrow(int32_t *h_dst, const int32_t *d_src, int64_t n, void *d_tmp_u16,
                           int32_t n_threads, void *stream);

/* ------------------------------------------------------------------------
 * EASE (`EASEScorer`, src/lenskit/knn/ease.py:88-147).
 * lk_ease_gram: the dense matrix the model inverts, G = X^T X + reg I for the binary
 *   users x items matrix X (`rates.co_occurrences("item", include_self=True, dense=True)` +
 *   the regularisation on the diagonal, ease.py:111-119; counting kernels
 *   src/accel/data/cooc.rs:47-192), from the off-diagonal co-occurrence counts in CSR form
 *   (the similarity build run on unit values, int64 offsets) and the item counts.
 * lk_ease_score_batch: `scores = q_vec @ self.weights` (ease.py:161-168) for a batch of
 *   queries: out[q][c] = sum over the history items i of query q of weights[i][c]
 *   (hist_items[hist_ptr[q]..hist_ptr[q+1]); items outside [0, n_items) are skipped).
 * ---------------------------------------------------------------------- */
int lk_ease_gram(const int64_t *d_cooc_indptr, const int32_t *d_cooc_indices,
                 const float *d_cooc_values, const int32_t *d_item_counts, int64_t n_items,
                 float reg, float *d_out, int64_t ld_out, void *stream);
int lk_ease_score_batch(const int64_t *d_hist_ptr, const int32_t *d_hist_items,
                        int64_t n_queries, const float *d_weights, int64_t n_items, int64_t ld_w,
                        float *d_out, int64_t ld_out, void *stream);

/* Batched fold-in (new-user embeddings) -- `ImplicitMFScorer.new_user_embedding` /
 * `_train_new_row` (src/lenskit/als/_implicit.py:77-130) -- is the SAME algebra as one ALS
 * row with OtOr = Q^T Q + user_reg I: build a plan over the histories' CSR offsets and call
 * lk_als_implicit_half_epoch with `this` = the [n_queries x ld] output and `other` = Q. */

/* ------------------------------------------------------------------------
 * Stable CSR transpose on the device.
 * Replaces `_accel.data.transpose_csr(structure, permute)` (src/lenskit/_accel/data.pyi:12,
 * src/accel/data/transpose.rs:19-108; `SparseRowArray.transpose`, data/matrix.py:512-530):
 * out_indptr [n_cols+1] (same width as the input offsets), out_indices [nnz] = source rows,
 * out_perm [nnz] (same width; NULL = not wanted) = source entry position of every output
 * entry; inside an output row the input's entry order is kept (counting sort), so the
 * result equals the reference's entry for entry.  Asynchronous on `stream`.
 * ---------------------------------------------------------------------- */
size_t lk_csr_transpose_workspace_bytes(int64_t nnz, int64_t n_cols, int indptr_is_64);
int lk_csr_transpose(const void *d_indptr, int indptr_is_64, const int32_t *d_indices,
                     int64_t n_rows, int64_t n_cols, int64_t nnz, void *d_out_indptr,
                     int32_t *d_out_indices, void *d_out_perm, void *d_ws, size_t ws_bytes,
                     void *stream);

/* Relabelling of a CSR matrix on the device (set-up of the sharded ALS engine; the reference
 * builds both orientations on the host with SciPy, src/lenskit/als/_common.py:216-222):
 * output row r takes the entries of input row d_row_src[r] (-1: an empty padding row), every
 * column c becomes d_col_map[c]; the entry order inside a row is kept (NOT re-sorted by the new
 * column numbers: the row solve sums over a row's entries, any order).  d_out_indptr holds the
 * offsets of the result (same width as the input's), computed by the caller from the row
 * lengths; d_values / d_out_values may be NULL (structure only).  Asynchronous on `stream`. */
int lk_csr_relabel(const void *d_indptr, int indptr_is_64, const int32_t *d_indices,
                   const float *d_values, int64_t n_rows_out, const int32_t *d_row_src,
                   const void *d_out_indptr, const int32_t *d_col_map, int32_t *d_out_indices,
                   float *d_out_values, void *stream);

/* Row gather of a CSR matrix on the device: the histories of a BATCH of queries cut out of the
 * training matrix in one launch.  Replaces, for a whole batch, the per-query
 * `UserTrainingHistoryLookup.__call__` -> `MatrixRelationshipSet.row_items`
 * (src/lenskit/basic/history.py:37-95) + the value vector of `ImplicitMFScorer.new_user_embedding`
 * (src/lenskit/als/_implicit.py:77-99: `ratings * weight`, or `np.full(n, weight)`), which the
 * reference's batch runner repeats query by query (src/lenskit/batch/_runner.py:283-308).
 * Output row r takes the entries of input row d_rows[r] (-1: an empty row) in their stored order;
 * d_out_indptr [n_rows_out + 1] (int64, the caller's prefix sums of the picked rows' lengths)
 * says where; out value = (d_values[e] - d_col_bias[column]) * scale in float32 (d_values NULL:
 * 1.0f in its place; d_col_bias NULL: nothing subtracted -- the item-kNN scorer mean-centres
 * the history ratings this way, src/lenskit/knn/item.py:268-271; d_out_values NULL: structure
 * only).  Asynchronous on `stream`. */
int lk_csr_gather_rows(const void *d_indptr, int indptr_is_64, const int32_t *d_indices,
                       const float *d_values, int64_t n_rows_out, const int32_t *d_rows,
                       const int64_t *d_out_indptr, const float *d_col_bias, float scale,
                       int32_t *d_out_indices, float *d_out_values, void *stream);

/* ------------------------------------------------------------------------
 * Item-kNN rating normalisation (`ItemKNNScorer._center_ratings` / `_normalize_rows`,
 * src/lenskit/knn/item.py:202-228) on the ITEM-MAJOR matrix produced by lk_csr_transpose:
 *   lk_iknn_prep_center: c = r - means[item] (d_means NULL: implicit feedback, c = r),
 *       d_sumsq[item] = the |c|^2 of the item added one by one in entry (= user) order --
 *       the order of the reference's norm -- and *d_nonzero_flag = any |c| > 1e-8;
 *   lk_iknn_prep_scale: value = c * d_recip[item], written item-major and, through the
 *       transpose's permutation (same width as the offsets), user-major.
 * The per-item means (np.add.reduceat sums / counts) and recip = 1 / max(sqrt(sumsq),
 * FLT_MIN) are computed by the caller with the reference's own NumPy calls, which makes the
 * whole preparation bit-identical to the reference's.
 * ---------------------------------------------------------------------- */
int lk_iknn_prep_center(const void *d_item_indptr, int indptr_is_64, const float *d_item_values,
                        const float *d_means, int64_t n_items, float *d_centered,
                        float *d_sumsq, int32_t *d_nonzero_flag, void *stream);
int lk_iknn_prep_scale(const void *d_item_indptr, int indptr_is_64, const void *d_perm,
                       const float *d_centered, const float *d_recip, int64_t n_items,
                       float *d_item_values_out, float *d_user_values_out, void *stream);

/* ------------------------------------------------------------------------
 * Synthetic interaction matrix in HBM (bench tooling; SURVEY.md section 8d "cfg5 concrete
 * input": Philox-seeded, shard-wise on the device -- the reference has no counterpart, its
 * benchmarks read MovieLens files).  Given the row offsets d_indptr [n_rows+1] (int64; the
 * degrees are the caller's), row r receives exactly its degree DISTINCT items in ascending
 * order, item rank drawn ~ Zipf(1.0) from Philox4x32-10(key = seed, counter = (r, j)) and
 * duplicates pushed to the next free rank; any row range can be generated independently.
 * Rows longer than 32 entries (at most 4096) must be listed in d_long_rows. */
int lk_synth_zipf_rows(const int64_t *d_indptr, int64_t n_rows, int64_t n_items, uint64_t seed,
                       const int32_t *d_long_rows, int64_t n_long_rows, int32_t *d_out_indices,
                       void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LKAMD_H */
