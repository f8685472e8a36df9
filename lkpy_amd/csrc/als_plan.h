// als_plan.h -- the ALS plan object shared by the Cholesky and CG half-epoch kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>

#define LK_DELTA_BLOCKS 256

// LK_ALS_WB64_K128: longest row (32 or 64 entries; 0 = none) that the 32 x 32 / 64 x 64 Woodbury
// systems of als_wb64_kernel take at padded k = 128 (at k = 256: always up to 64)
#ifndef LK_ALS_WB64_K128_DEFAULT
#define LK_ALS_WB64_K128_DEFAULT 64
#endif
static inline int wb64_k128_limit()
{
    const char *e = getenv("LK_ALS_WB64_K128");
    return e ? atoi(e) : LK_ALS_WB64_K128_DEFAULT;
}

#ifndef LK_ALS_CHUNK
#define LK_ALS_CHUNK 1024  // CSR entries per chunk of a long row
#endif
#ifndef LK_ALS_LONG_ROW
#define LK_ALS_LONG_ROW 2048  // rows longer than this are chunked
#endif
#define LK_ALS_CHUNK_BLK LK_ALS_CHUNK  // same chunking for the workgroup-per-row kernel (als_blk.hip)
// Rows with more than LK_ALS_SLAB_GROUP chunks get their slabs pre-summed in groups of that many
// (slab_group_reduce_kernel: every group in parallel, chunk order inside a group), and the solve
// kernel adds the group heads only.  Two short float32 sums instead of one long one: the
// busiest cfg5 item (1.54 M entries = 1505 slabs) was 1.07e-4 from the float64 solution with the
// single sequential sum; and the 209 MB of its slabs are no longer read by ONE workgroup.
#ifndef LK_ALS_SLAB_GROUP
#define LK_ALS_SLAB_GROUP 16
#endif

struct lk_als_plan {
    int64_t n_rows = 0;
    int32_t k = 0, KP = 0, NT = 0, solver = 0;
    int32_t is64 = 0;  // width of the CSR offsets the plan was built from
    int64_t n_chunks = 0;
    int64_t n_long = 0;
    // rows with <= 16 entries are a suffix [t_short, n_rows) of the longest-first order; for
    // padded k > 64 the implicit model solves them through the Woodbury kernel (als_wb.hip)
    // when the caller supplied Z = other * OtOr^-1 for this half-epoch (lk_als_plan_set_z)
    int64_t t_short = 0;
    int64_t t_mid = 0;  // rows with 17 .. 64 entries are [t_mid, t_short): als_wb64_kernel
    int64_t t_4 = 0;    // rows with <= 4 entries are [t_4, n_rows): als_wb4_kernel, four per wave
    int64_t t_8 = 0;    // rows with 5 .. 8 entries are [t_8, t_4): als_wb4_kernel<.., 8>, two per wave
    int64_t t_32 = 0;   // rows with 17 .. 32 entries are [t_32, t_short): the 32 x 32 system of als_wb64_kernel (KP = 128)
    int64_t t_128 = 0;  // rows with 65 .. 128 entries are [t_128, t_mid): als_wb128_kernel (KP = 256)
    // rows [t_cg, n_rows) have at most 16384 / KP entries (256 / 128 / 64 at padded k = 64 /
    // 128 / 256): what the CG kernel keeps in registers over its iterations (als_cg.hip)
    int64_t t_cg = 0;
    int64_t t_cg1 = 0;  // rows [t_cg1, n_rows): at most 4096 / KP entries -- one wave's registers
    mutable const float *d_z = nullptr;
    // ... or a caller-owned [n_cols x KP] buffer the LIBRARY fills with Z at every implicit
    // half-epoch (lk_als_plan_set_z_workspace): OtOr^-1 by spd_inverse.hip, Z by the scoring GEMM
    float *d_zbuf = nullptr;
    // ... or Z as ANOTHER plan of the same half-epoch forms it (lk_als_plan_set_z_shared: the row
    // slices of a sharded half-epoch share one Z): d_z = that plan's buffer, d_zflag_src = the
    // device word holding its "OtOr is not positive definite" flag, copied into this plan's
    // status[1] at every launch
    const int *d_zflag_src = nullptr;
    bool z_for_others = false;  // leading slice: form Z even if this slice has no short row itself
    // optional [n_rows x KP] buffer (lk_als_plan_set_rhs_workspace): every half-epoch first forms
    // the right-hand side y in the REFERENCE's order (als_rhs.hip: one sequential float32 chain
    // per feature, implicit.rs:116-117) and the dense solve kernels take it from there
    float *d_yref = nullptr;
    // LK_ALS_PLAN_REFERENCE_ORDER (lk_als_plan_create_ex): the normal matrix is summed in the
    // reference's order as well -- matrixmultiply's KC = 256 blocks (lk_oracle.c, lko_gram_mtl_m):
    // every row with more than 256 entries is cut into 256-entry chunks, each chunk one MFMA fmaf
    // chain from zero, the chunk slabs added ONE AFTER THE OTHER in chunk order (one slab group
    // per row), OtOr added last.  Such a plan needs the rhs workspace (d_yref) to run.
    bool ref_order = false;
    // LK_ALS_PLAN_HYBRID_ORDER (the default of lk_als_plan_create): the same two sums in the
    // reference's order, but ONLY for the rows longer than `long_row` entries (LK_ALS_REF_LEN,
    // default 2048) -- exactly the rows that are pre-reduced in chunks anyway: 256-entry chunks, slabs
    // added in chunk order, OtOr last, y from the sequential chain of als_rhs.hip (kept in the
    // plan's own workspace, one row of KP floats per long row, indexed by TASK: the long rows are
    // the first n_long tasks of the longest-first order).  Shorter rows keep the tuned kernels'
    // own order, where the two arithmetics agree to ~1e-5.  Works with task control, row offsets,
    // relabelled / sharded engines (the order of a row's entries is whatever the CSR holds).
    bool hybrid = false;
    size_t off_yref = 0;                 // hybrid: [n_long x KP] floats in the workspace
    int32_t chunk = LK_ALS_CHUNK;        // CSR entries per chunk (= per slab) of a long row
    // entries per WORK UNIT of the chunk kernel (a multiple of `chunk`): hybrid plans at padded
    // k = 64 (als_chunk_kernel) and k = 256 (the LDS-staged als_blk_chunk_dma_kernel) keep the
    // tuned kernels' 1024-entry units -- one wave / workgroup runs its gather ring across the four
    // 256-entry blocks of a unit and stores a slab at every block boundary (d_chunk_slab = the
    // unit's first slab); everywhere else a unit is one chunk
    int32_t unit = LK_ALS_CHUNK;
    int64_t n_slabs = 0;                 // slabs of all long rows (n_chunks counts the UNITS)
    int32_t *d_chunk_slab = nullptr;     // [n_chunks] first slab of the unit
    int32_t long_row = LK_ALS_LONG_ROW;  // rows longer than this are chunked
    size_t off_ginv = 0, off_invws = 0;  // [KP x KP] float inverse, spd_inverse scratch
    // device-side schedule
    char *d_pack = nullptr;          // the one device allocation the arrays below live in
    size_t pack_cap = 0;             // its size (it may come from, and return to, a pool)
    int32_t *d_order = nullptr;      // [n_rows] rows, longest first
    int32_t *d_row_slab = nullptr;   // [n_rows] first slab of the row or -1
    int32_t *d_chunk_row = nullptr;  // [n_chunks]
    int64_t *d_chunk_beg = nullptr;  // [n_chunks] CSR entry range
    int32_t *d_chunk_len = nullptr;
    int64_t n_groups = 0;           // slab groups of the rows with more than LK_ALS_SLAB_GROUP chunks
    int32_t *d_grp_head = nullptr;  // [n_groups] first slab of the group
    int32_t *d_grp_cnt = nullptr;   // [n_groups] slabs in the group (2 .. LK_ALS_SLAB_GROUP)
    // workspace layout (byte offsets)
    size_t off_status = 0, off_otor = 0, off_delta = 0, off_partial = 0, off_slabs = 0,
           ws_bytes = 0;
    // >= 0 (set by the CG half-epoch around its call of the exact launchers): solve only the
    // rows [0, dense_limit) of the order -- the chunked rows -- and leave the rest to the caller
    mutable int64_t dense_limit = -1;
    struct lk_task_ctl *ctl = nullptr;  // optional cancel / progress block (lk_als_plan_set_ctl)
    float cg_tol = 1e-7f;
    int32_t cg_max_iter = 0;
    // optional per-kernel timing (HIP events on the launch stream): ring of
    // (start, mid, stop) triples -- start..mid = chunk kernel, mid..stop = solve kernel
    static constexpr int TIMING_RING = 128;
    bool timing = false;
    mutable int timing_n = 0;
    mutable hipEvent_t ev[TIMING_RING][3] = {};
    // side stream of the half-epoch: OtOr^-1 (spd_inverse.hip: a ONE-workgroup sweep, 0.25 ms at
    // k = 128, 1.7 ms at k = 256) runs there under the chunk kernel instead of in front of the
    // Z GEMM on the launch stream (created on first use; LK_ALS_SIDE_STREAM=0: launch stream)
    mutable hipStream_t side = nullptr;
    mutable hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // second side stream: the sequential right-hand-side chains of the long rows (als_rhs.hip)
    // run there beside the chunk kernel and the solve of the short rows; only the small solve
    // launch of the long rows waits for them (plan_fork_rhs / plan_join_rhs)
    mutable hipStream_t side_rhs = nullptr;
    mutable hipEvent_t ev_fork_rhs = nullptr, ev_join_rhs = nullptr, ev_mid_rhs = nullptr;
};

namespace lk {
// exact half-epoch for padded k <= 64: one wave per row (als_chol.hip; implicit model)
int als_chol_half_epoch(const lk_als_plan *p, const void *indptr, int is64, const int32_t *indices,
                        const float *values, int64_t n_rows, int k, float *this_, int ld_this,
                        const float *other, int ld_other, const float *otor, int ld_otor, char *ws,
                        float *out_frob, hipStream_t st);
// exact half-epoch for padded k = 128 / 256: one workgroup per row (als_blk.hip)
size_t als_blk_slab_floats(int NT);
int als_blk_half_epoch(const lk_als_plan *p, const void *indptr, int is64, const int32_t *indices,
                       const float *values, int64_t n_rows, int64_t n_cols, int k, float *this_,
                       const float *other, const float *otor, int ld_otor, char *ws,
                       float *out_frob, hipStream_t st, bool expl, float reg);
// exact half-epoch for padded k > 256 (multiples of 64 up to 1024): the blocked algorithm of
// als_blk.hip on tiles kept in an HBM scratch (als_big.hip)
size_t als_big_scratch_bytes(int KP, int64_t n_rows);
int als_big_half_epoch(const lk_als_plan *p, const void *indptr, int is64, const int32_t *indices,
                       const float *values, int64_t n_rows, int k, float *this_,
                       const float *other, const float *otor, int ld_otor, char *ws,
                       float *out_frob, hipStream_t st, bool expl, float reg);
size_t gramian_big_workspace_bytes(int KP);
int gramian_big(const float *m, int64_t n, int k, int KP, float reg, float *out, int ld_out,
                float *ws, hipStream_t st);
// rows [t0, n_rows) of the plan order (<= 16 entries each) through the Woodbury kernel (als_wb.hip)
int als_wb_launch(const lk_als_plan *p, const void *indptr, int is64, const int32_t *indices,
                  const float *values, int64_t t0, int64_t n_rows, float *this_,
                  const float *other, const float *z, float *row_delta, int *status,
                  hipStream_t st);
// rows [t0, n_rows) of the plan order (<= 4 entries each), four rows per wave (als_wb.hip)
int als_wb4_launch(const lk_als_plan *p, const void *indptr, int is64, const int32_t *indices,
                   const float *values, int64_t t0, int64_t n_rows, float *this_,
                   const float *other, const float *z, float *row_delta, int *status,
                   hipStream_t st, int slots = 4);
// (both Woodbury kernels return at once when status[1] != 0: Z is not available -- OtOr was not
// positive definite -- and the dense fallback launch of als_blk.hip solves their rows)
// rows [t0, t1) of the plan order (17 .. 64 entries each), als_wb64_kernel (als_chol.hip)
int als_wb64_launch(const lk_als_plan *p, const void *indptr, int is64, const int32_t *indices,
                    const float *values, int64_t t0, int64_t t1, float *this_,
                    const float *other, const float *z, float *row_delta, int *status,
                    hipStream_t st);
// out[KP x KP] = a[k x k]^-1 (float64 accuracy, zero padded), *flag != 0 when a is not positive
// definite (spd_inverse.hip); ws: spd_inverse_workspace_bytes(KP)
size_t spd_inverse_workspace_bytes(int KP);
int spd_inverse(const float *a, int lda, int k, int KP, float *out, int *flag, void *ws,
                hipStream_t st);
// y_out[t] = the right-hand side of row order[t] in the reference's summation order, for the
// tasks t in [0, n_tasks) (als_rhs.hip); y_out is [n_tasks x KP], natural feature order
int launch_rhs_reference(const lk_als_plan *p, const void *indptr, int is64,
                         const int32_t *indices, const float *values, const int32_t *order,
                         int64_t n_tasks, const float *other, bool expl, float *y_out,
                         hipStream_t st);
// the plan's rhs side stream (created on first use), forked behind what `st` holds now
// (LK_ALS_SIDE_STREAM=0: `st` itself); plan_join_rhs makes `st` wait for what it holds then
int plan_fork_rhs(const lk_als_plan *p, hipStream_t st, hipStream_t *side);
int plan_join_rhs(const lk_als_plan *p, hipStream_t st);
// the rhs side stream waits for what `st` holds now (a second fork point inside the half-epoch)
int plan_rhs_wait_main(const lk_als_plan *p, hipStream_t st);
// slab[head] += slab[head + 1] + ... (chunk order) for every group of the plan (als_chol.hip)
int launch_slab_group_reduce(const lk_als_plan *p, float *slabs, size_t slab_floats, hipStream_t st);
// deterministic two-stage sum of the per-row squared deltas -> sqrt (als_chol.hip)
int launch_delta_reduce(const float *row_delta, int64_t n_rows, float *partial, float *out_frob,
                        hipStream_t st);
}  // namespace lk
