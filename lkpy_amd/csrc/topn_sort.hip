// topn_sort.hip -- full descending sort of score rows (top-N with n = None / -1 / n > 4096).
//
// Stands in for `argsort_descending` / `argsort_float` (src/accel/data/sorting.rs:69-103),
// consumed by `ItemList.top_n(None)` (src/lenskit/data/_items.py:975-998, the `want_all`
// branch) and by `TopNRanker` without `n` (src/lenskit/basic/topn.py:61-69): ALL valid
// (non-NaN, non-null) entries of a row by descending score.  The reference's order among
// equal scores is unspecified (an unstable sort); here ties go to the LOWER index, the same
// rule as the top-N selection kernel, so the two paths agree on every prefix.
//
// Not a hot path of the benchmark (the ranking lists of the pipelines are n <= a few
// hundred and take the selection kernel of topk.hip): keys = (row << 32) | the INVERTED
// order-preserving unsigned image of the score (NaN -> key 0, inverted 0xffffffff: behind every
// valid key of its row), values = column numbers, one STABLE ascending radix sort of the 64-bit
// keys per batch of rows (radix_sort.h, this repository's own: rounds 1-5 called rocPRIM's
// segmented sort) -- rows stay where they are, inside a row the scores descend and equal scores
// keep the column order --, then the valid prefix of every row is emitted.  HBM traffic:
// 4 + ceil(log2(rows) / 8) passes x 32 B per entry.
#include "common.h"
#include "radix_sort.h"

namespace lk {

__global__ void sort_keys_kernel(const float *__restrict__ scores, int64_t ld_s, int64_t row_len,
                                 int64_t n_rows, unsigned long long *__restrict__ keys,
                                 uint32_t *__restrict__ vals)
{
    const int64_t total = n_rows * row_len;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / row_len;
        const int64_t c = e - r * row_len;
        const float x = scores[r * ld_s + c];
        const uint32_t k = (x == x) ? f2key(x) : 0u;  // every valid key is >= f2key(-inf) > 0
        keys[e] = ((unsigned long long)r << 32) | (unsigned long long)(0xffffffffu - k);
        vals[e] = (uint32_t)c;
    }
}

__global__ void sort_emit_kernel(const unsigned long long *__restrict__ keys,
                                 const uint32_t *__restrict__ vals, int64_t row_len,
                                 int64_t n_rows, int64_t n, int32_t *__restrict__ out_idx,
                                 float *__restrict__ out_score, int64_t out_ld)
{
    const int64_t total = n_rows * n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / n, j = e - r * n;
        uint32_t k = 0u;
        int32_t v = -1;
        if (j < row_len) {
            k = 0xffffffffu - (uint32_t)(keys[r * row_len + j] & 0xffffffffull);
            v = (int32_t)vals[r * row_len + j];
        }
        const bool ok = k != 0u;
        out_idx[r * out_ld + j] = ok ? v : -1;
        if (out_score) out_score[r * out_ld + j] = ok ? key2f(k) : __builtin_nanf("");
    }
}

// rows sorted per call: bounded so that offsets fit 32 bits and the workspace stays modest
static int64_t sort_batch_rows(int64_t n_rows, int64_t row_len)
{
    const int64_t cap = ((int64_t)1 << 28) / (row_len > 0 ? row_len : 1);  // 2^28 entries
    int64_t b = cap < 1 ? 1 : cap;
    return b < n_rows ? b : n_rows;
}

static int row_bits(int64_t rows)
{
    int b = 0;
    while (((int64_t)1 << b) < rows) ++b;
    return b;
}

size_t topn_sort_workspace_bytes(int64_t n_rows, int64_t row_len)
{
    if (n_rows <= 0 || row_len <= 0) return 256;
    const int64_t b = sort_batch_rows(n_rows, row_len);
    const size_t e = (size_t)b * (size_t)row_len;
    // keys in / out / ping-pong (8 B), values in / out / ping-pong (4 B), the sort's histogram
    return 3 * align_up(e * 8, 256) + 3 * align_up(e * 4, 256) +
           align_up(radix_sort_temp_bytes((int64_t)e), 256) + 256;
}

// rows of `scores` (row stride ld_s) -> out_idx[r][0..n) (and out_score): every valid entry
// by descending score, ties by lower column, padded with -1 / NaN.
int topn_sort(const float *scores, int64_t ld_s, int64_t n_rows, int64_t row_len, int64_t n,
              void *ws, int32_t *out_idx, float *out_score, int64_t out_ld, hipStream_t st)
{
    if (n_rows <= 0 || n <= 0) return LK_OK;
    LK_REQUIRE(row_len < ((int64_t)1 << 31), "top-N sort: rows longer than 2^31 entries");
    const int64_t batch = sort_batch_rows(n_rows, row_len);
    const size_t e = (size_t)batch * (size_t)row_len;
    char *p = static_cast<char *>(ws);
    auto *k_in = reinterpret_cast<unsigned long long *>(p);
    p += align_up(e * 8, 256);
    auto *k_out = reinterpret_cast<unsigned long long *>(p);
    p += align_up(e * 8, 256);
    auto *k_tmp = reinterpret_cast<unsigned long long *>(p);
    p += align_up(e * 8, 256);
    uint32_t *v_in = reinterpret_cast<uint32_t *>(p);
    p += align_up(e * 4, 256);
    uint32_t *v_out = reinterpret_cast<uint32_t *>(p);
    p += align_up(e * 4, 256);
    uint32_t *v_tmp = reinterpret_cast<uint32_t *>(p);
    p += align_up(e * 4, 256);
    void *tmp = p;
    for (int64_t r0 = 0; r0 < n_rows; r0 += batch) {
        const int64_t rows = (n_rows - r0) < batch ? (n_rows - r0) : batch;
        const int64_t tot = rows * row_len;
        if (row_len > 0) {
            const unsigned g = (unsigned)((tot + 255) / 256 < 8192 ? (tot + 255) / 256 : 8192);
            hipLaunchKernelGGL(sort_keys_kernel, dim3(g ? g : 1), dim3(256), 0, st,
                               scores + r0 * ld_s, ld_s, row_len, rows, k_in, v_in);
            int rc = radix_sort_pairs<unsigned long long, uint32_t>(
                k_in, v_in, k_out, v_out, k_tmp, v_tmp, tot, 0, 32 + row_bits(rows), tmp, st);
            if (rc != LK_OK) return rc;
        }
        const int64_t ot = rows * n;
        const unsigned g2 = (unsigned)((ot + 255) / 256 < 8192 ? (ot + 255) / 256 : 8192);
        hipLaunchKernelGGL(sort_emit_kernel, dim3(g2 ? g2 : 1), dim3(256), 0, st, k_out, v_out,
                           row_len, rows, n, out_idx + r0 * out_ld,
                           out_score ? out_score + r0 * out_ld : nullptr, out_ld);
    }
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

}  // namespace lk
