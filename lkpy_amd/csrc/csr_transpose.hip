// csr_transpose.hip -- stable CSR transpose on the device, gfx950.
//
// Stands in for `transpose_csr` / `transpose_structure` (src/accel/data/transpose.rs:19-108;
// used by `SparseRowArray.transpose`, src/lenskit/data/matrix.py:512-530): a counting sort of
// the entries by column that keeps, inside each output row (= input column), the input's
// entry order (= ascending input row).  Outputs: offsets of the transposed matrix, its
// column indices (= input rows) and, optionally, the permutation (input entry position of
// every output entry) with which values are carried over.
//
// A stable LSD radix sort of (column -> entry position) IS that counting sort, so the
// result is identical to the reference's, entry for entry (integer work: bit-exact).  The
// sort is this repository's own (radix_sort.h: 8-bit digits, stable passes, tiles reordered
// through LDS; rounds 1-5 called rocPRIM here); the offsets come from a binary search per
// output row over the sorted columns, the source rows from a binary search per entry over the
// input offsets.  HBM bound: (12 + 2 W) * nnz bytes per pass over ceil(log2(n_cols) / 8) passes
// (W = 4 or 8: the width of an entry position).
#include <cstring>

#include "common.h"
#include "radix_sort.h"

namespace lk {

template <typename VT>
__global__ void tr_iota_kernel(VT *__restrict__ v, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        v[i] = (VT)i;
}

// out_ptr[c] = first position in the sorted column list with column >= c
template <typename IT>
__global__ void tr_offsets_kernel(const uint32_t *__restrict__ sorted_cols, int64_t nnz,
                                  int64_t n_cols, IT *__restrict__ out_ptr)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_cols) return;
    int64_t lo = 0, hi = nnz;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)sorted_cols[mid] < c)
            lo = mid + 1;
        else
            hi = mid;
    }
    out_ptr[c] = (IT)lo;
}

// out_idx[pos] = row containing input entry perm[pos]; out_perm[pos] = perm[pos]
template <typename IT, typename VT>
__global__ void tr_rows_kernel(const IT *__restrict__ in_ptr, int64_t n_rows,
                               const VT *__restrict__ perm, int64_t nnz,
                               int32_t *__restrict__ out_idx, IT *__restrict__ out_perm)
{
    for (int64_t pos = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pos < nnz;
         pos += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = (int64_t)perm[pos];
        int64_t lo = 0, hi = n_rows;  // last row with in_ptr[row] <= e
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)in_ptr[mid + 1] <= e)
                lo = mid + 1;
            else
                hi = mid;
        }
        out_idx[pos] = (int32_t)lo;
        if (out_perm) out_perm[pos] = (IT)e;
    }
}

// relabelled matrix: output row r = input row row_src[r] (or empty), columns mapped; one wave
// per output row, entries copied in order
template <typename IT>
__global__ __launch_bounds__(256) void relabel_kernel(
    const IT *__restrict__ in_ptr, const int32_t *__restrict__ in_idx,
    const float *__restrict__ in_val, int64_t n_rows_out, const int32_t *__restrict__ row_src,
    const IT *__restrict__ out_ptr, const int32_t *__restrict__ col_map,
    int32_t *__restrict__ out_idx, float *__restrict__ out_val)
{
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows_out) return;
    const int src = row_src[r];
    if (src < 0) return;
    const int lane = threadIdx.x & 63;
    const int64_t sb = (int64_t)in_ptr[src], n = (int64_t)in_ptr[src + 1] - sb;
    const int64_t db = (int64_t)out_ptr[r];
    for (int64_t e = lane; e < n; e += 64) {
        out_idx[db + e] = col_map[in_idx[sb + e]];
        if (out_val) out_val[db + e] = in_val[sb + e];
    }
}

// row gather: output row r = input row rows[r] (or empty), indices copied, values copied and
// scaled (or the constant `scale` where the matrix holds no values); one wave per output row
template <typename IT>
__global__ __launch_bounds__(256) void gather_rows_kernel(
    const IT *__restrict__ in_ptr, const int32_t *__restrict__ in_idx,
    const float *__restrict__ in_val, int64_t n_rows_out, const int32_t *__restrict__ rows,
    const int64_t *__restrict__ out_ptr, const float *__restrict__ col_bias, float scale,
    int32_t *__restrict__ out_idx, float *__restrict__ out_val)
{
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows_out) return;
    const int src = rows[r];
    if (src < 0) return;
    const int lane = threadIdx.x & 63;
    const int64_t sb = (int64_t)in_ptr[src];
    const int64_t db = out_ptr[r], n = out_ptr[r + 1] - db;  // (the caller's lengths decide)
    for (int64_t e = lane; e < n; e += 64) {
        const int32_t c = in_idx[sb + e];
        out_idx[db + e] = c;
        if (out_val) {
            float v = in_val ? in_val[sb + e] : 1.0f;
            if (col_bias) v -= col_bias[c];
            out_val[db + e] = v * scale;
        }
    }
}

static int key_bits(int64_t n_cols)
{
    int b = 1;
    while (b < 32 && ((int64_t)1 << b) < n_cols) ++b;
    return b;
}

template <typename IT, typename VT>
static int transpose_impl(const IT *in_ptr, const int32_t *in_idx, int64_t n_rows, int64_t n_cols,
                          int64_t nnz, IT *out_ptr, int32_t *out_idx, IT *out_perm, char *ws,
                          size_t ws_bytes, hipStream_t st)
{
    // workspace: sorted columns | column ping-pong | positions (iota) | sorted positions |
    // position ping-pong | the sort's histogram
    uint32_t *keys_out = reinterpret_cast<uint32_t *>(ws);
    size_t off = align_up((size_t)nnz * 4, 256);
    uint32_t *keys_tmp = reinterpret_cast<uint32_t *>(ws + off);
    off += align_up((size_t)nnz * 4, 256);
    VT *vals_in = reinterpret_cast<VT *>(ws + off);
    off += align_up((size_t)nnz * sizeof(VT), 256);
    VT *vals_out = reinterpret_cast<VT *>(ws + off);
    off += align_up((size_t)nnz * sizeof(VT), 256);
    VT *vals_tmp = reinterpret_cast<VT *>(ws + off);
    off += align_up((size_t)nnz * sizeof(VT), 256);
    void *tmp = ws + off;
    (void)ws_bytes;
    const uint32_t *keys_in = reinterpret_cast<const uint32_t *>(in_idx);
    if (nnz > 0) {
        hipLaunchKernelGGL(tr_iota_kernel<VT>, dim3(1024), dim3(256), 0, st, vals_in, nnz);
        int rc = radix_sort_pairs<uint32_t, VT>(keys_in, vals_in, keys_out, vals_out, keys_tmp,
                                                vals_tmp, nnz, 0, key_bits(n_cols), tmp, st);
        if (rc != LK_OK) return rc;
        hipLaunchKernelGGL((tr_rows_kernel<IT, VT>), dim3(2048), dim3(256), 0, st, in_ptr, n_rows,
                           vals_out, nnz, out_idx, out_perm);
    }
    hipLaunchKernelGGL(tr_offsets_kernel<IT>, dim3((unsigned)((n_cols + 256) / 256)), dim3(256),
                       0, st, keys_out, nnz, n_cols, out_ptr);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

}  // namespace lk

extern "C" size_t lk_csr_transpose_workspace_bytes(int64_t nnz, int64_t n_cols, int indptr_is_64)
{
    if (nnz < 0 || n_cols < 0) return 0;
    const size_t vt = indptr_is_64 ? 8 : 4;
    (void)n_cols;
    return 2 * lk::align_up((size_t)nnz * 4, 256) + 3 * lk::align_up((size_t)nnz * vt, 256) +
           lk::align_up(lk::radix_sort_temp_bytes(nnz), 256) + 256;
}

extern "C" int lk_csr_transpose(const void *d_indptr, int indptr_is_64, const int32_t *d_indices,
                                int64_t n_rows, int64_t n_cols, int64_t nnz, void *d_out_indptr,
                                int32_t *d_out_indices, void *d_out_perm, void *d_ws,
                                size_t ws_bytes, void *stream)
{
    LK_REQUIRE(n_rows >= 0 && n_cols >= 0 && nnz >= 0, "lk_csr_transpose: negative size");
    LK_REQUIRE(n_rows < (int64_t)INT32_MAX && n_cols < (int64_t)INT32_MAX,
               "lk_csr_transpose: dimensions must fit int32 indices");
    LK_REQUIRE(indptr_is_64 || nnz < (int64_t)INT32_MAX,
               "lk_csr_transpose: nnz needs 64-bit offsets");
    LK_REQUIRE(d_indptr && d_out_indptr && d_ws && (nnz == 0 || (d_indices && d_out_indices)),
               "lk_csr_transpose: null pointer");
    LK_REQUIRE(ws_bytes >= lk_csr_transpose_workspace_bytes(nnz, n_cols, indptr_is_64),
               "lk_csr_transpose: workspace too small");
    hipStream_t st = lk::as_stream(stream);
    char *ws = static_cast<char *>(d_ws);
    if (indptr_is_64)
        return lk::transpose_impl<int64_t, uint64_t>(
            static_cast<const int64_t *>(d_indptr), d_indices, n_rows, n_cols, nnz,
            static_cast<int64_t *>(d_out_indptr), d_out_indices,
            static_cast<int64_t *>(d_out_perm), ws, ws_bytes, st);
    return lk::transpose_impl<int32_t, uint32_t>(
        static_cast<const int32_t *>(d_indptr), d_indices, n_rows, n_cols, nnz,
        static_cast<int32_t *>(d_out_indptr), d_out_indices, static_cast<int32_t *>(d_out_perm),
        ws, ws_bytes, st);
}

extern "C" int lk_csr_relabel(const void *d_indptr, int indptr_is_64, const int32_t *d_indices,
                              const float *d_values, int64_t n_rows_out, const int32_t *d_row_src,
                              const void *d_out_indptr, const int32_t *d_col_map,
                              int32_t *d_out_indices, float *d_out_values, void *stream)
{
    LK_REQUIRE(n_rows_out >= 0, "lk_csr_relabel: negative size");
    if (n_rows_out == 0) return LK_OK;
    LK_REQUIRE(d_indptr && d_row_src && d_out_indptr && d_col_map, "lk_csr_relabel: null pointer");
    hipStream_t st = lk::as_stream(stream);
    const dim3 grid((unsigned)((n_rows_out + 3) / 4)), block(256);
    if (indptr_is_64)
        hipLaunchKernelGGL(lk::relabel_kernel<int64_t>, grid, block, 0, st,
                           static_cast<const int64_t *>(d_indptr), d_indices, d_values, n_rows_out,
                           d_row_src, static_cast<const int64_t *>(d_out_indptr), d_col_map,
                           d_out_indices, d_out_values);
    else
        hipLaunchKernelGGL(lk::relabel_kernel<int32_t>, grid, block, 0, st,
                           static_cast<const int32_t *>(d_indptr), d_indices, d_values, n_rows_out,
                           d_row_src, static_cast<const int32_t *>(d_out_indptr), d_col_map,
                           d_out_indices, d_out_values);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

extern "C" int lk_csr_gather_rows(const void *d_indptr, int indptr_is_64, const int32_t *d_indices,
                                  const float *d_values, int64_t n_rows_out, const int32_t *d_rows,
                                  const int64_t *d_out_indptr, const float *d_col_bias, float scale,
                                  int32_t *d_out_indices, float *d_out_values, void *stream)
{
    LK_REQUIRE(n_rows_out >= 0, "lk_csr_gather_rows: negative size");
    if (n_rows_out == 0) return LK_OK;
    LK_REQUIRE(d_indptr && d_rows && d_out_indptr, "lk_csr_gather_rows: null pointer");
    hipStream_t st = lk::as_stream(stream);
    const dim3 grid((unsigned)((n_rows_out + 3) / 4)), block(256);
    if (indptr_is_64)
        hipLaunchKernelGGL(lk::gather_rows_kernel<int64_t>, grid, block, 0, st,
                           static_cast<const int64_t *>(d_indptr), d_indices, d_values, n_rows_out,
                           d_rows, d_out_indptr, d_col_bias, scale, d_out_indices, d_out_values);
    else
        hipLaunchKernelGGL(lk::gather_rows_kernel<int32_t>, grid, block, 0, st,
                           static_cast<const int32_t *>(d_indptr), d_indices, d_values, n_rows_out,
                           d_rows, d_out_indptr, d_col_bias, scale, d_out_indices, d_out_values);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}
