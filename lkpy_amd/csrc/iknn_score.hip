// iknn_score.hip -- item-kNN scoring for a BATCH of queries, gfx950.
//
// Stands in for `score_explicit` / `score_implicit` (src/accel/knn/item_score.rs:23-111)
// and `ScoreAccumulator` (src/accel/knn/accum.rs:16-239), which score ONE query per call
// on one CPU thread (the reference's "batch" is a Python loop,
// src/lenskit/batch/_runner.py:283-308).
//
// Per query: for every reference (history) item r, in history order, its similarity
// row S_r is scattered into per-target accumulators that keep the `max_nbrs` largest
// similarities (an entry only displaces the current minimum when STRICTLY larger,
// accum.rs:108); score_t = sum(s*v)/sum(s) (explicit) or sum(s) (implicit) over the kept
// entries, NaN when fewer than `min_nbrs`; count_t = entries kept.  Null (negative)
// reference items are skipped (the reference would read out of bounds, SURVEY.md section 8a
// row a14); null targets give NaN / -1.  A NaN similarity is an error
// (accum.rs:146-151 -> ValueError("similarity is null")).
//
// One workgroup per query.  History items are taken one at a time (a similarity row has
// distinct columns, so the threads of the workgroup never collide inside one row and the
// per-target state is updated with plain loads/stores; a barrier separates rows).  The
// per-target slots live in a per-workgroup slab of the workspace (L2-resident):
// cnt[t] (-1 = not a target), minval[t], and max_nbrs (s, v) slots.
#include "common.h"

namespace lk {

constexpr int KS_THREADS = 512;
constexpr int KS_MAX_WGS = 256;

struct KsSlab {
    int32_t *cnt;
    float *minval;
    float *slot_s;
    float *slot_v;
};

__host__ __device__ inline size_t ks_slab_bytes(int64_t n_items, int max_nbrs)
{
    size_t per = (size_t)n_items * (sizeof(int32_t) + sizeof(float)) +
                 (size_t)n_items * max_nbrs * 2 * sizeof(float);
    return (per + 255) / 256 * 256;
}

// SWAP = false: item-kNN (matrix = similarities, entry value = weight, the reference row's
// rating = value).  SWAP = true: user-kNN (`user_score_items_*`,
// src/accel/knn/user_score.rs:21-98): matrix = the users' ratings, the reference rows are the
// neighbours, their similarity (ref_rates) = weight, the entry value (rating; none for
// implicit feedback: s_val null) = value.  `n_rows` = rows of the matrix, `n_items` = columns.
template <bool SWAP>
__global__ __launch_bounds__(KS_THREADS) void iknn_score_kernel(
    const int64_t *__restrict__ s_ptr, const int32_t *__restrict__ s_idx,
    const float *__restrict__ s_val, int64_t n_rows, int64_t n_items, int64_t n_queries,
    const int64_t *__restrict__ ref_ptr, const int32_t *__restrict__ ref_items,
    const float *__restrict__ ref_rates, const int64_t *__restrict__ tgt_ptr,
    const int32_t *__restrict__ tgt_items, int max_nbrs, int min_nbrs, char *__restrict__ ws,
    size_t slab_bytes, float *__restrict__ out_scores, int32_t *__restrict__ out_counts,
    int *__restrict__ status)
{
    char *slab = ws + (size_t)blockIdx.x * slab_bytes;
    int32_t *cnt = reinterpret_cast<int32_t *>(slab);
    float *minval = reinterpret_cast<float *>(slab + (size_t)n_items * 4);
    float *slot_s = reinterpret_cast<float *>(slab + (size_t)n_items * 8);
    float *slot_v = slot_s + (size_t)n_items * max_nbrs;
    const int tid = threadIdx.x;
    const bool explicit_ = SWAP ? (s_val != nullptr) : (ref_rates != nullptr);

    for (int64_t q = blockIdx.x; q < n_queries; q += gridDim.x) {
        const int64_t rb = ref_ptr[q], re = ref_ptr[q + 1];
        const int64_t tb = tgt_ptr[q], te = tgt_ptr[q + 1];
        // enable the targets (ScoreAccumulator::new_array, accum.rs:30-37)
        for (int64_t t = tid; t < n_items; t += KS_THREADS) cnt[t] = -1;
        __syncthreads();
        for (int64_t j = tb + tid; j < te; j += KS_THREADS) {
            const int t = tgt_items[j];
            if (t >= 0 && t < n_items) cnt[t] = 0;
        }
        __syncthreads();
        // scatter the history rows, one reference item at a time, in history order
        for (int64_t r = rb; r < re; ++r) {
            const int ri = ref_items[r];
            if (ri < 0 || ri >= n_rows) continue;  // wave-uniform: null reference row
            const float rscalar = (SWAP || explicit_) ? ref_rates[r] : 0.f;
            const int64_t sb = s_ptr[ri], se = s_ptr[ri + 1];
            for (int64_t e = sb + tid; e < se; e += KS_THREADS) {
                const int t = s_idx[e];
                const int c = cnt[t];
                if (c < 0) continue;
                // (weight, value) of this contribution
                const float s = SWAP ? rscalar : s_val[e];
                const float rv = SWAP ? (explicit_ ? s_val[e] : 0.f) : rscalar;
                if (s != s) {
                    atomicCAS(status, 0, 1);
                    continue;
                }
                float *ss = slot_s + (size_t)t * max_nbrs;
                float *sv = slot_v + (size_t)t * max_nbrs;
                if (c < max_nbrs) {  // Partial(vec): push (accum.rs:103-104)
                    ss[c] = s;
                    sv[c] = rv;
                    cnt[t] = c + 1;
                    if (c + 1 == max_nbrs) {  // becomes Full: cache the minimum
                        float m = ss[0];
                        for (int i = 1; i < max_nbrs; ++i) m = fminf(m, ss[i]);
                        minval[t] = m;
                    }
                } else if (s > minval[t]) {  // Full: displace the minimum (accum.rs:106-113)
                    int mi = 0;
                    float m = ss[0];
                    for (int i = 1; i < max_nbrs; ++i)
                        if (ss[i] < m) {
                            m = ss[i];
                            mi = i;
                        }
                    ss[mi] = s;
                    sv[mi] = rv;
                    m = ss[0];
                    for (int i = 1; i < max_nbrs; ++i) m = fminf(m, ss[i]);
                    minval[t] = m;
                }
            }
            __syncthreads();
        }
        // collect (collect_items_averaged / _summed / _counts, accum.rs:186-239)
        for (int64_t j = tb + tid; j < te; j += KS_THREADS) {
            const int t = tgt_items[j];
            float score = __builtin_nanf("");
            int count = -1;
            if (t >= 0 && t < n_items) {
                const int c = cnt[t];
                count = c;
                if (c >= min_nbrs && c > 0) {
                    const float *ss = slot_s + (size_t)t * max_nbrs;
                    const float *sv = slot_v + (size_t)t * max_nbrs;
                    float tw = 0.f, wsum = 0.f;
                    for (int i = 0; i < c; ++i) {
                        tw += ss[i];
                        wsum += ss[i] * sv[i];
                    }
                    score = explicit_ ? wsum / tw : tw;
                }
            }
            out_scores[j] = score;
            out_counts[j] = count;
        }
        __syncthreads();
    }
}

}  // namespace lk

extern "C" size_t lk_iknn_score_workspace_bytes(int64_t n_items, int64_t n_queries,
                                                int32_t max_nbrs)
{
    if (n_items < 0 || max_nbrs < 1) return 0;
    int64_t wgs = n_queries < lk::KS_MAX_WGS ? n_queries : lk::KS_MAX_WGS;
    if (wgs < 1) wgs = 1;
    return 256 + (size_t)wgs * lk::ks_slab_bytes(n_items, max_nbrs);
}

extern "C" int lk_iknn_score_batch(const int64_t *d_sim_indptr, const int32_t *d_sim_indices,
                                   const float *d_sim_values, int64_t n_items, int64_t n_queries,
                                   const int64_t *d_ref_ptr, const int32_t *d_ref_items,
                                   const float *d_ref_rates, const int64_t *d_tgt_ptr,
                                   const int32_t *d_tgt_items, int32_t max_nbrs, int32_t min_nbrs,
                                   void *d_ws, float *d_out_scores, int32_t *d_out_counts,
                                   void *stream)
{
    LK_REQUIRE(max_nbrs >= 1 && min_nbrs >= 1, "lk_iknn_score_batch: max_nbrs/min_nbrs must be >= 1");
    LK_REQUIRE(n_items >= 0 && n_queries >= 0, "lk_iknn_score_batch: negative size");
    if (n_queries == 0) return LK_OK;
    LK_REQUIRE(d_sim_indptr && d_ref_ptr && d_tgt_ptr && d_ws && d_out_scores && d_out_counts,
               "lk_iknn_score_batch: null pointer");
    hipStream_t st = lk::as_stream(stream);
    char *ws = static_cast<char *>(d_ws);
    int *status = reinterpret_cast<int *>(ws);
    LK_HIP_CHECK(hipMemsetAsync(status, 0, 256, st));
    int64_t wgs = n_queries < lk::KS_MAX_WGS ? n_queries : lk::KS_MAX_WGS;
    hipLaunchKernelGGL(lk::iknn_score_kernel<false>, dim3((unsigned)wgs), dim3(lk::KS_THREADS), 0,
                       st, d_sim_indptr, d_sim_indices, d_sim_values, n_items, n_items, n_queries,
                       d_ref_ptr,
                       d_ref_items, d_ref_rates, d_tgt_ptr, d_tgt_items, max_nbrs, min_nbrs,
                       ws + 256, lk::ks_slab_bytes(n_items, max_nbrs), d_out_scores, d_out_counts,
                       status);
    LK_HIP_CHECK(hipGetLastError());
    int h = 0;
    LK_HIP_CHECK(hipMemcpyAsync(&h, status, sizeof(int), hipMemcpyDeviceToHost, st));
    LK_HIP_CHECK(hipStreamSynchronize(st));
    if (h != 0) {
        lk::set_error("similarity is null");
        return LK_E_NAN_SIM;
    }
    return LK_OK;
}

extern "C" int lk_uknn_score_batch(const int64_t *d_rat_indptr, const int32_t *d_rat_indices,
                                   const float *d_rat_values, int64_t n_users, int64_t n_items,
                                   int64_t n_queries, const int64_t *d_nbr_ptr,
                                   const int32_t *d_nbr_rows, const float *d_nbr_sims,
                                   const int64_t *d_tgt_ptr, const int32_t *d_tgt_items,
                                   int32_t max_nbrs, int32_t min_nbrs, void *d_ws,
                                   float *d_out_scores, int32_t *d_out_counts, void *stream)
{
    LK_REQUIRE(max_nbrs >= 1 && min_nbrs >= 1, "lk_uknn_score_batch: max_nbrs/min_nbrs must be >= 1");
    LK_REQUIRE(n_users >= 0 && n_items >= 0 && n_queries >= 0, "lk_uknn_score_batch: negative size");
    if (n_queries == 0) return LK_OK;
    LK_REQUIRE(d_rat_indptr && d_nbr_ptr && d_tgt_ptr && d_ws && d_out_scores && d_out_counts,
               "lk_uknn_score_batch: null pointer");
    hipStream_t st = lk::as_stream(stream);
    char *ws = static_cast<char *>(d_ws);
    int *status = reinterpret_cast<int *>(ws);
    LK_HIP_CHECK(hipMemsetAsync(status, 0, 256, st));
    int64_t wgs = n_queries < lk::KS_MAX_WGS ? n_queries : lk::KS_MAX_WGS;
    hipLaunchKernelGGL(lk::iknn_score_kernel<true>, dim3((unsigned)wgs), dim3(lk::KS_THREADS), 0,
                       st, d_rat_indptr, d_rat_indices, d_rat_values, n_users, n_items, n_queries,
                       d_nbr_ptr, d_nbr_rows, d_nbr_sims, d_tgt_ptr, d_tgt_items, max_nbrs,
                       min_nbrs, ws + 256, lk::ks_slab_bytes(n_items, max_nbrs), d_out_scores,
                       d_out_counts, status);
    LK_HIP_CHECK(hipGetLastError());
    int h = 0;
    LK_HIP_CHECK(hipMemcpyAsync(&h, status, sizeof(int), hipMemcpyDeviceToHost, st));
    LK_HIP_CHECK(hipStreamSynchronize(st));
    if (h != 0) {
        lk::set_error("similarity is null");
        return LK_E_NAN_SIM;
    }
    return LK_OK;
}

// ---------------------------------------------------------------------------
// Neighbour similarities of user-kNN: sims[q][u] = <user_vectors[u], x_q> for a batch of dense
// query vectors (`nbr_sims = self.user_vectors @ ratings`, src/lenskit/knn/user.py:196): one
// wave per matrix row, lane = query (blocks of 64 queries), the row's entries broadcast.
// x is [n_items x ld_x] (item-major: the queries of one item are contiguous).
// ---------------------------------------------------------------------------
namespace lk {
template <bool IS64>
__global__ __launch_bounds__(256) void csr_rows_dot_kernel(
    const typename IndPtr<IS64>::type *__restrict__ ptr, const int32_t *__restrict__ idx,
    const float *__restrict__ val, int64_t n_rows, const float *__restrict__ x, int64_t ld_x,
    int64_t n_queries, float *__restrict__ out, int64_t ld_out)
{
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const int64_t b = ptr[row], e = ptr[row + 1];
    for (int64_t q0 = 0; q0 < n_queries; q0 += 64) {
        const int64_t q = q0 + lane;
        float acc = 0.f;
        for (int64_t base = b; base < e; base += 64) {
            // 64 entries of the row per coalesced load, then broadcast one by one
            const int64_t me = base + lane;
            const int it = me < e ? idx[me] : 0;
            const float v = me < e ? val[me] : 0.f;
            const int n = (e - base) < 64 ? (int)(e - base) : 64;
            for (int j = 0; j < n; ++j) {
                const int itj = __shfl(it, j, 64);
                const float vj = __shfl(v, j, 64);
                if (q < n_queries) acc = fmaf(vj, x[(int64_t)itj * ld_x + q], acc);
            }
        }
        if (q < n_queries) out[q * ld_out + row] = acc;
    }
}
}  // namespace lk

extern "C" int lk_csr_rows_dot(const void *d_indptr, int indptr_is_64, const int32_t *d_indices,
                               const float *d_values, int64_t n_rows, const float *d_x,
                               int64_t ld_x, int64_t n_queries, float *d_out, int64_t ld_out,
                               void *stream)
{
    LK_REQUIRE(n_rows >= 0 && n_queries >= 0 && ld_x >= n_queries && ld_out >= n_rows,
               "lk_csr_rows_dot: bad shape");
    if (n_rows == 0 || n_queries == 0) return LK_OK;
    LK_REQUIRE(d_indptr && d_x && d_out, "lk_csr_rows_dot: null pointer");
    hipStream_t st = lk::as_stream(stream);
    const dim3 grid((unsigned)((n_rows + 3) / 4)), block(256);
    if (indptr_is_64)
        hipLaunchKernelGGL(lk::csr_rows_dot_kernel<true>, grid, block, 0, st,
                           static_cast<const int64_t *>(d_indptr), d_indices, d_values, n_rows,
                           d_x, ld_x, n_queries, d_out, ld_out);
    else
        hipLaunchKernelGGL(lk::csr_rows_dot_kernel<false>, grid, block, 0, st,
                           static_cast<const int32_t *>(d_indptr), d_indices, d_values, n_rows,
                           d_x, ld_x, n_queries, d_out, ld_out);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

// ---------------------------------------------------------------------------
// EASE (src/lenskit/knn/ease.py).
//
// lk_ease_gram: the dense, regularised co-occurrence Gramian the model inverts,
//     G = X^T X + reg I   (X = the binary users x items matrix; ease.py:111-119),
// from the device similarity build run on unit values (off-diagonal cells: every shared user
// adds 1.0f, exact integers) plus the item counts on the diagonal.  One lane per stored cell.
//
// lk_ease_score_batch: scores = q_vec @ weights (ease.py:161-168) for a batch of queries:
// q_vec is the 0/1 indicator of the query's history, so a query's scores are the SUM of the
// history items' weight rows.  Workgroup = (query, 256-column strip), rows added in history
// order (plain f32 adds: deterministic).
// ---------------------------------------------------------------------------
namespace lk {
__global__ void ease_gram_fill_kernel(const int64_t *__restrict__ ptr,
                                      const int32_t *__restrict__ idx,
                                      const float *__restrict__ val, int64_t n,
                                      float *__restrict__ g, int64_t ld)
{
    const int64_t row = blockIdx.x;
    const int64_t b = ptr[row], e = ptr[row + 1];
    for (int64_t i = b + threadIdx.x; i < e; i += blockDim.x) g[row * ld + idx[i]] = val[i];
}

__global__ void ease_gram_diag_kernel(const int32_t *__restrict__ counts, int64_t n, float reg,
                                      float *__restrict__ g, int64_t ld)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) g[i * ld + i] = (float)counts[i] + reg;
}

__global__ __launch_bounds__(256) void ease_score_kernel(
    const int64_t *__restrict__ hist_ptr, const int32_t *__restrict__ hist_items,
    const float *__restrict__ w, int64_t n_items, int64_t ld_w, float *__restrict__ out,
    int64_t ld_out)
{
    const int64_t q = blockIdx.y;
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n_items) return;
    const int64_t b = hist_ptr[q], e = hist_ptr[q + 1];
    float acc = 0.f;
    for (int64_t i = b; i < e; ++i) {
        const int32_t it = hist_items[i];
        if (it >= 0 && it < n_items) acc += w[(int64_t)it * ld_w + c];
    }
    out[q * ld_out + c] = acc;
}
}  // namespace lk

extern "C" int lk_ease_gram(const int64_t *d_cooc_indptr, const int32_t *d_cooc_indices,
                            const float *d_cooc_values, const int32_t *d_item_counts,
                            int64_t n_items, float reg, float *d_out, int64_t ld_out, void *stream)
{
    LK_REQUIRE(n_items >= 0 && ld_out >= n_items, "lk_ease_gram: bad shape");
    if (n_items == 0) return LK_OK;
    LK_REQUIRE(d_cooc_indptr && d_item_counts && d_out, "lk_ease_gram: null pointer");
    hipStream_t st = lk::as_stream(stream);
    LK_HIP_CHECK(hipMemsetAsync(d_out, 0, (size_t)n_items * ld_out * sizeof(float), st));
    hipLaunchKernelGGL(lk::ease_gram_fill_kernel, dim3((unsigned)n_items), dim3(256), 0, st,
                       d_cooc_indptr, d_cooc_indices, d_cooc_values, n_items, d_out, ld_out);
    hipLaunchKernelGGL(lk::ease_gram_diag_kernel, dim3((unsigned)((n_items + 255) / 256)),
                       dim3(256), 0, st, d_item_counts, n_items, reg, d_out, ld_out);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

extern "C" int lk_ease_score_batch(const int64_t *d_hist_ptr, const int32_t *d_hist_items,
                                   int64_t n_queries, const float *d_weights, int64_t n_items,
                                   int64_t ld_w, float *d_out, int64_t ld_out, void *stream)
{
    LK_REQUIRE(n_queries >= 0 && n_items >= 0 && ld_w >= n_items && ld_out >= n_items,
               "lk_ease_score_batch: bad shape");
    if (n_queries == 0 || n_items == 0) return LK_OK;
    LK_REQUIRE(d_hist_ptr && d_weights && d_out, "lk_ease_score_batch: null pointer");
    LK_REQUIRE(n_queries <= 65535, "lk_ease_score_batch: at most 65535 queries per call");
    hipLaunchKernelGGL(lk::ease_score_kernel,
                       dim3((unsigned)((n_items + 255) / 256), (unsigned)n_queries), dim3(256), 0,
                       lk::as_stream(stream), d_hist_ptr, d_hist_items, d_weights, n_items, ld_w,
                       d_out, ld_out);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}
