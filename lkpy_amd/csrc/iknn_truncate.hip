// iknn_truncate.hip -- per-row top-`save_nbrs` truncation of the similarity matrix, gfx950.
//
// Second half of `sim_row` (src/accel/knn/item_train.rs:139-151): when `save_nbrs` is set,
// the kept neighbours of a row are stable-sorted by similarity descending, truncated to
// `save_nbrs`, and re-sorted by column.  "Stable" means ties keep the order of FIRST
// ENCOUNTER (`used`, item_train.rs:112-126): by the first user the two items share (users
// are walked in ascending order), then by column (a user's items are walked in ascending
// order).  Only ties AT the cut matter, so:
//
//   1. MSB-first radix select finds the save_nbrs-th largest similarity of the row
//      (similarities are positive, so the float bit pattern orders like an unsigned int);
//   2. everything above it is kept; if more entries EQUAL it than there is room for, each
//      tied entry gets its first common user (two sorted-list intersection on the item x
//      user CSR) and the tied entries with the smallest (first user, column) are kept --
//      two more radix selects;
//   3. a compaction pass writes the kept entries in column order.
//
// One workgroup per row; the full (untruncated) matrix comes from the build kernel and
// stays in HBM.  Bit-exact index sets by construction.
#include "common.h"

namespace lk {

constexpr int TR_THREADS = 256;

// block-wide: the `need`-th SMALLEST (ascending) or LARGEST (descending) key among the
// entries e in [0, n) with pred(e); returns the key; *n_take_eq = how many entries equal to
// it are still needed.  key_of(e) must be cheap (it is evaluated 5 times).
template <bool LARGEST, typename KeyFn, typename PredFn>
__device__ unsigned block_radix_select(int64_t n, unsigned need, KeyFn key_of, PredFn pred,
                                       unsigned *hist, unsigned *s_prefix, unsigned *s_need)
{
    const int tid = threadIdx.x;
    if (tid == 0) {
        *s_prefix = 0;
        *s_need = need;
    }
    __syncthreads();
    for (int shift = 24; shift >= 0; shift -= 8) {
        hist[tid] = 0;
        __syncthreads();
        const unsigned prefix = *s_prefix;
        const unsigned hmask = (shift == 24) ? 0u : (0xffffffffu << (shift + 8));
        for (int64_t e = tid; e < n; e += TR_THREADS) {
            if (pred(e)) {
                const unsigned k = key_of(e);
                if ((k & hmask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
            }
        }
        __syncthreads();
        if (tid == 0) {
            unsigned nd = *s_need, cum = 0;
            int b;
            if (LARGEST) {
                for (b = 255; b > 0; --b) {
                    if (cum + hist[b] >= nd) break;
                    cum += hist[b];
                }
            } else {
                for (b = 0; b < 255; ++b) {
                    if (cum + hist[b] >= nd) break;
                    cum += hist[b];
                }
            }
            *s_need = nd - cum;
            *s_prefix = prefix | ((unsigned)b << shift);
        }
        __syncthreads();
    }
    return *s_prefix;
}

// first user shared by items a and b (both rows of the item x user CSR are sorted)
template <bool IS64>
__device__ int first_common_user(const typename IndPtr<IS64>::type *__restrict__ iu_ptr,
                                 const int32_t *__restrict__ iu_idx, int a, int b)
{
    int64_t pa = iu_ptr[a], ea = iu_ptr[a + 1];
    int64_t pb = iu_ptr[b], eb = iu_ptr[b + 1];
    while (pa < ea && pb < eb) {
        const int ua = iu_idx[pa], ub = iu_idx[pb];
        if (ua == ub) return ua;
        if (ua < ub) {  // gallop a forward to the first user >= ub
            int64_t lo = pa + 1, hi = ea;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (iu_idx[mid] < ub)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            pa = lo;
        } else {
            int64_t lo = pb + 1, hi = eb;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (iu_idx[mid] < ua)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            pb = lo;
        }
    }
    return 0x7fffffff;  // cannot happen for a positive similarity
}

template <bool IS64>
__global__ __launch_bounds__(TR_THREADS) void iknn_trunc_select_kernel(
    const int64_t *__restrict__ s_ptr, const int32_t *__restrict__ s_idx,
    const float *__restrict__ s_val, const typename IndPtr<IS64>::type *__restrict__ iu_ptr,
    const int32_t *__restrict__ iu_idx, int64_t row_begin, int save_nbrs,
    uint8_t *__restrict__ keep, int32_t *__restrict__ first_u, int32_t *__restrict__ new_cnt)
{
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_need, s_cnt;
    const int row = blockIdx.x;
    const int tid = threadIdx.x;
    const int64_t b = s_ptr[row];
    const int64_t n = s_ptr[row + 1] - b;
    const int32_t *idx = s_idx + b;
    const float *val = s_val + b;
    uint8_t *kp = keep + b;
    if (n <= save_nbrs) {  // short row: everything stays (item_train.rs:145 truncate is a no-op)
        for (int64_t e = tid; e < n; e += TR_THREADS) kp[e] = 1;
        if (tid == 0) new_cnt[row] = (int32_t)n;
        return;
    }
    auto vkey = [&](int64_t e) { return __builtin_bit_cast(unsigned, val[e]); };
    // Lower bound of the save_nbrs-th largest similarity: the save_nbrs-th largest of the
    // 256 per-thread maxima (at least save_nbrs entries are >= it).  The radix select then
    // only histograms the few entries above the bound -- per-element LDS atomics are the
    // expensive part of it (~3 cycles per element per CU).
    unsigned tau = 0;
    if (save_nbrs <= TR_THREADS) {
        __shared__ unsigned tmax[TR_THREADS];
        unsigned best = 0;
        for (int64_t e = tid; e < n; e += TR_THREADS) best = max(best, vkey(e));
        tmax[tid] = best;
        __syncthreads();
        for (unsigned k = 2; k <= TR_THREADS; k <<= 1) {
            for (unsigned j = k >> 1; j > 0; j >>= 1) {
                const unsigned i = tid, ixj = i ^ j;
                if (ixj > i) {
                    const unsigned a = tmax[i], c = tmax[ixj];
                    const bool desc = ((i & k) == 0);
                    if (desc ? (a < c) : (a > c)) {
                        tmax[i] = c;
                        tmax[ixj] = a;
                    }
                }
                __syncthreads();
            }
        }
        tau = tmax[save_nbrs - 1];  // > 0: n > save_nbrs entries, all positive
    }
    auto above = [&](int64_t e) { return vkey(e) >= tau; };
    const unsigned kth = block_radix_select<true>(n, (unsigned)save_nbrs, vkey, above, hist,
                                                  &s_prefix, &s_need);
    const unsigned need_eq = s_need;
    __syncthreads();
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    {
        unsigned c = 0;
        for (int64_t e = tid; e < n; e += TR_THREADS) {
            const unsigned k = vkey(e);
            kp[e] = (k > kth) ? 1 : 0;
            c += (k == kth) ? 1u : 0u;
        }
        atomicAdd(&s_cnt, c);
    }
    __syncthreads();
    const unsigned n_eq = s_cnt;
    if (n_eq == need_eq) {  // no tie across the cut
        for (int64_t e = tid; e < n; e += TR_THREADS)
            if (vkey(e) == kth) kp[e] = 1;
    } else {
        // ties: order of first encounter = (first common user, column)
        int32_t *fu = first_u + b;
        for (int64_t e = tid; e < n; e += TR_THREADS)
            if (vkey(e) == kth)
                fu[e] = first_common_user<IS64>(iu_ptr, iu_idx, (int)row_begin + row, idx[e]);
        __syncthreads();
        auto tied = [&](int64_t e) { return vkey(e) == kth; };
        auto ukey = [&](int64_t e) { return (unsigned)fu[e]; };
        const unsigned u_th = block_radix_select<false>(n, need_eq, ukey, tied, hist, &s_prefix,
                                                        &s_need);
        const unsigned need_u = s_need;  // entries with first user == u_th still needed
        __syncthreads();
        auto tied_u = [&](int64_t e) { return vkey(e) == kth && (unsigned)fu[e] == u_th; };
        auto ckey = [&](int64_t e) { return (unsigned)idx[e]; };
        const unsigned c_th = block_radix_select<false>(n, need_u, ckey, tied_u, hist, &s_prefix,
                                                        &s_need);
        __syncthreads();
        for (int64_t e = tid; e < n; e += TR_THREADS) {
            if (vkey(e) == kth) {
                const unsigned u = (unsigned)fu[e];
                // columns are unique inside a row, so (u, column) is a total order
                if (u < u_th || (u == u_th && (unsigned)idx[e] <= c_th)) kp[e] = 1;
            }
        }
    }
    if (tid == 0) new_cnt[row] = save_nbrs;
}

// exclusive scan int32 -> int64 (n + 1 outputs), one workgroup
__global__ __launch_bounds__(1024) void trunc_scan_kernel(const int32_t *__restrict__ cnt,
                                                         int64_t n, int64_t *__restrict__ off)
{
    __shared__ int64_t wsum[16];
    __shared__ int64_t carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const int64_t v = (i < n) ? (int64_t)cnt[i] : 0;
        int64_t x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int64_t y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int64_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const int64_t carry = carry_s;
        if (i < n) off[i] = carry + woff + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) off[n] = carry_s;
}

// kept entries of a row, in their (column) order, to the row's new position
__global__ __launch_bounds__(TR_THREADS) void iknn_trunc_compact_kernel(
    const int64_t *__restrict__ s_ptr, const int32_t *__restrict__ s_idx,
    const float *__restrict__ s_val, const uint8_t *__restrict__ keep,
    const int64_t *__restrict__ new_ptr, int32_t *__restrict__ out_idx,
    float *__restrict__ out_val)
{
    __shared__ unsigned wcnt[TR_THREADS / 64];
    __shared__ unsigned s_base;
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t b = s_ptr[row], n = s_ptr[row + 1] - b;
    const int64_t ob = new_ptr[row];
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int64_t e0 = 0; e0 < n; e0 += TR_THREADS) {
        const int64_t e = e0 + tid;
        const bool k = (e < n) && keep[b + e] != 0;
        const unsigned long long m = __ballot(k);
        if (lane == 0) wcnt[w] = (unsigned)__popcll(m);
        __syncthreads();
        unsigned before = s_base;
        for (int q = 0; q < w; ++q) before += wcnt[q];
        if (k) {
            const unsigned pos = before + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            out_idx[ob + pos] = s_idx[b + e];
            out_val[ob + pos] = s_val[b + e];
        }
        __syncthreads();
        if (tid == 0) {
            unsigned t = 0;
            for (int q = 0; q < TR_THREADS / 64; ++q) t += wcnt[q];
            s_base += t;
        }
        __syncthreads();
    }
}

}  // namespace lk

extern "C" size_t lk_iknn_truncate_workspace_bytes(int64_t n_items, int64_t nnz)
{
    if (n_items < 0 || nnz < 0) return 0;
    return lk::align_up((size_t)nnz, 256) + lk::align_up((size_t)nnz * 4, 256) +
           lk::align_up((size_t)(n_items + 1) * 4, 256) + 256;
}

extern "C" int lk_iknn_truncate_count(const int64_t *d_sim_indptr, const int32_t *d_sim_indices,
                                      const float *d_sim_values, const void *d_iu_indptr,
                                      int iu_indptr_is_64, const int32_t *d_iu_indices,
                                      int64_t n_items, int64_t row_begin, int64_t nnz,
                                      int64_t save_nbrs, void *d_ws, int64_t *d_out_indptr,
                                      int64_t *h_total_nnz, void *stream)
{
    LK_REQUIRE(save_nbrs > 0 && save_nbrs < (int64_t)INT32_MAX,
               "lk_iknn_truncate_count: save_nbrs must be positive");
    LK_REQUIRE(n_items >= 0 && nnz >= 0 && row_begin >= 0,
               "lk_iknn_truncate_count: negative size");
    LK_REQUIRE(d_sim_indptr && d_ws && d_out_indptr && h_total_nnz && d_iu_indptr,
               "lk_iknn_truncate_count: null pointer");
    hipStream_t st = lk::as_stream(stream);
    char *ws = static_cast<char *>(d_ws);
    uint8_t *keep = reinterpret_cast<uint8_t *>(ws);
    int32_t *first_u = reinterpret_cast<int32_t *>(ws + lk::align_up((size_t)nnz, 256));
    int32_t *cnt = reinterpret_cast<int32_t *>(ws + lk::align_up((size_t)nnz, 256) +
                                               lk::align_up((size_t)nnz * 4, 256));
    if (n_items > 0) {
        if (iu_indptr_is_64)
            hipLaunchKernelGGL((lk::iknn_trunc_select_kernel<true>), dim3((unsigned)n_items),
                               dim3(lk::TR_THREADS), 0, st, d_sim_indptr, d_sim_indices,
                               d_sim_values, static_cast<const int64_t *>(d_iu_indptr),
                               d_iu_indices, row_begin, (int)save_nbrs, keep, first_u, cnt);
        else
            hipLaunchKernelGGL((lk::iknn_trunc_select_kernel<false>), dim3((unsigned)n_items),
                               dim3(lk::TR_THREADS), 0, st, d_sim_indptr, d_sim_indices,
                               d_sim_values, static_cast<const int32_t *>(d_iu_indptr),
                               d_iu_indices, row_begin, (int)save_nbrs, keep, first_u, cnt);
    }
    hipLaunchKernelGGL(lk::trunc_scan_kernel, dim3(1), dim3(1024), 0, st, cnt, n_items,
                       d_out_indptr);
    LK_HIP_CHECK(hipGetLastError());
    LK_HIP_CHECK(hipMemcpyAsync(h_total_nnz, d_out_indptr + n_items, sizeof(int64_t),
                                hipMemcpyDeviceToHost, st));
    LK_HIP_CHECK(hipStreamSynchronize(st));
    return LK_OK;
}

extern "C" int lk_iknn_truncate_fill(const int64_t *d_sim_indptr, const int32_t *d_sim_indices,
                                     const float *d_sim_values, int64_t n_items, int64_t nnz,
                                     void *d_ws, const int64_t *d_out_indptr,
                                     int32_t *d_out_indices, float *d_out_values, void *stream)
{
    LK_REQUIRE(n_items >= 0 && nnz >= 0, "lk_iknn_truncate_fill: negative size");
    if (n_items == 0) return LK_OK;
    LK_REQUIRE(d_sim_indptr && d_ws && d_out_indptr, "lk_iknn_truncate_fill: null pointer");
    const uint8_t *keep = static_cast<const uint8_t *>(d_ws);
    hipLaunchKernelGGL(lk::iknn_trunc_compact_kernel, dim3((unsigned)n_items),
                       dim3(lk::TR_THREADS), 0, lk::as_stream(stream), d_sim_indptr,
                       d_sim_indices, d_sim_values, keep, d_out_indptr, d_out_indices,
                       d_out_values);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}
