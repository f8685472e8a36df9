// synth.hip -- seeded synthetic interaction matrices generated in HBM (bench tooling).
//
// SURVEY.md section 8d, "cfg5 concrete input": U = 10^7 users, I = 10^6 items, nnz = 10^8,
// per-user degree ~ truncated Zipf (mean 10, min 1), items chosen proportionally to Zipf(1.0),
// "generated shard-wise on device from seed (Philox)".  The reference has no such generator
// (its benchmarks read MovieLens files); this only manufactures the roofline workload without
// a 10^8-entry host round trip.
//
// Given the row offsets (the degrees are drawn on the host: 10^7 numbers), every row is filled
// independently -- any shard of rows can be generated on any rank from (seed, row) alone:
//   rank_j = floor(I^u_j) - 1 clipped to [0, I),  u_j = Philox4x32-10(key = seed, ctr = (row, j))
// (the inverse CDF of the continuous 1/x density: Zipf(1.0) up to discretisation); the d ranks
// are sorted and made strictly increasing, r'_j = j + max_{i<=j}(r_i - i) (duplicates are pushed
// to the next free rank; an overflow past I-1 is pushed back from the top), so a row has
// exactly its d DISTINCT items, ascending.  Item id = rank (popular items have small ids).
// Short rows (<= 32): one thread per row, insertion sort in registers/scratch; longer rows (up
// to LK_SYNTH_MAX_ROW = 4096): one wave per row, bitonic sort + prefix-max scan in LDS.
#include "common.h"

#define LK_SYNTH_MAX_ROW 4096

namespace lk {

struct Philox {
    uint32_t k0, k1;
    __device__ __forceinline__ void round(uint32_t (&c)[4], uint32_t ka, uint32_t kb) const
    {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ ka;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ kb;
        c[1] = (uint32_t)p1;
        c[3] = (uint32_t)p0;
        c[0] = n0;
        c[2] = n2;
    }
    // Philox4x32-10: 128-bit counter -> 4 x 32 random bits
    __device__ __forceinline__ void operator()(uint32_t (&c)[4]) const
    {
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            round(c, a, b);
            a += 0x9E3779B9u;
            b += 0xBB67AE85u;
        }
    }
};

// rank in [0, n_items) with P(rank = r) ~ 1 / (r + 1)
__device__ __forceinline__ int zipf_rank(uint64_t seed, int64_t row, uint32_t j, int64_t n_items,
                                         float log_n)
{
    Philox ph{(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t c[4] = {(uint32_t)row, (uint32_t)((uint64_t)row >> 32), j, 0x5eedu};
    ph(c);
    const float u = ((c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0, 1)
    const float x = __expf(u * log_n);                            // [1, n_items + 1)
    int64_t r = (int64_t)x - 1;
    if (r < 0) r = 0;
    if (r >= n_items) r = n_items - 1;
    return (int)r;
}

// rows of <= 32 entries: one thread per row
__global__ void synth_short_rows_kernel(const int64_t *__restrict__ indptr, int64_t n_rows,
                                        int64_t n_items, uint64_t seed, float log_n,
                                        int32_t *__restrict__ out)
{
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    const int64_t b = indptr[row];
    const int d = (int)(indptr[row + 1] - b);
    if (d <= 0 || d > 32) return;
    int r[32];
    for (int j = 0; j < d; ++j) {  // insertion sort
        const int v = zipf_rank(seed, row, (uint32_t)j, n_items, log_n);
        int p = j;
        while (p > 0 && r[p - 1] > v) {
            r[p] = r[p - 1];
            --p;
        }
        r[p] = v;
    }
    // strictly increasing: r'_j = j + max_{i <= j} (r_i - i); then cap from the top
    int m = -(1 << 30);
    for (int j = 0; j < d; ++j) {
        m = max(m, r[j] - j);
        r[j] = j + m;
    }
    int cap = (int)n_items - 1;
    for (int j = d - 1; j >= 0; --j) {
        r[j] = min(r[j], cap);
        cap = r[j] - 1;
    }
    for (int j = 0; j < d; ++j) out[b + j] = r[j];
}

// rows of 33 .. LK_SYNTH_MAX_ROW entries: one wave per row (listed in `rows`)
__global__ __launch_bounds__(64) void synth_long_rows_kernel(
    const int64_t *__restrict__ indptr, const int32_t *__restrict__ rows, int64_t n_sel,
    int64_t n_items, uint64_t seed, float log_n, int32_t *__restrict__ out)
{
    __shared__ int key[LK_SYNTH_MAX_ROW];
    const int lane = threadIdx.x;
    if ((int64_t)blockIdx.x >= n_sel) return;
    const int64_t row = rows[blockIdx.x];
    const int64_t b = indptr[row];
    const int d = (int)(indptr[row + 1] - b);
    int p2 = 64;
    while (p2 < d) p2 <<= 1;
    for (int j = lane; j < p2; j += 64)
        key[j] = j < d ? zipf_rank(seed, row, (uint32_t)j, n_items, log_n) : 0x7fffffff;
    __syncthreads();
    for (int k = 2; k <= p2; k <<= 1)
        for (int s = k >> 1; s > 0; s >>= 1) {
            for (int i = lane; i < p2; i += 64) {
                const int x = i ^ s;
                if (x > i) {
                    const int a = key[i], c = key[x];
                    const bool up = (i & k) == 0;
                    if (up ? (a > c) : (a < c)) {
                        key[i] = c;
                        key[x] = a;
                    }
                }
            }
            __syncthreads();
        }
    // prefix max of (r_i - i): blocked over the wave, carried sequentially
    int carry = -(1 << 30);
    for (int base = 0; base < d; base += 64) {
        const int j = base + lane;
        int v = j < d ? key[j] - j : -(1 << 30);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(v, o, 64);
            if (lane >= o) v = max(v, t);
        }
        v = max(v, carry);
        if (j < d) key[j] = j + v;
        carry = __shfl(v, 63, 64);
    }
    __syncthreads();
    // cap from the top: r_j <= n_items - 1 - (d - 1 - j)
    for (int j = lane; j < d; j += 64) {
        const int lim = (int)n_items - 1 - (d - 1 - j);
        out[b + j] = min(key[j], lim);
    }
}

}  // namespace lk

extern "C" int lk_synth_zipf_rows(const int64_t *d_indptr, int64_t n_rows, int64_t n_items,
                                  uint64_t seed, const int32_t *d_long_rows, int64_t n_long_rows,
                                  int32_t *d_out_indices, void *stream)
{
    LK_REQUIRE(n_rows >= 0 && n_items >= 1 && n_items < (int64_t)INT32_MAX,
               "lk_synth_zipf_rows: bad shape");
    if (n_rows == 0) return LK_OK;
    LK_REQUIRE(d_indptr && d_out_indices, "lk_synth_zipf_rows: null pointer");
    hipStream_t st = lk::as_stream(stream);
    const float log_n = logf((float)n_items + 1.0f);
    hipLaunchKernelGGL(lk::synth_short_rows_kernel, dim3((unsigned)((n_rows + 255) / 256)),
                       dim3(256), 0, st, d_indptr, n_rows, n_items, seed, log_n, d_out_indices);
    if (n_long_rows > 0) {
        LK_REQUIRE(d_long_rows, "lk_synth_zipf_rows: null long-row list");
        hipLaunchKernelGGL(lk::synth_long_rows_kernel, dim3((unsigned)n_long_rows), dim3(64), 0,
                           st, d_indptr, d_long_rows, n_long_rows, n_items, seed, log_n,
                           d_out_indices);
    }
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}
