// gramian.hip -- out = M^T M + reg*I on gfx950 (f32 MFMA, deterministic).
//
// Stands in for `_implicit_otor` (src/lenskit/als/_implicit.py:177-184), which is a
// NumPy sgemm in the reference.  M is [n x ld] row-major with ld = padded k and
// zero pad columns.
//
// Layout / mapping
//   The k features are handled in "primed" order: feature f = s*NT + t  <->
//   primed index p = t*16 + s  (t = 16-wide tile, s = lane & 15), so that one
//   lane loads its NT features of a row as ONE contiguous NT-float vector and a
//   16-lane group covers the whole row with a single coalesced request.
//   v_mfma_f32_16x16x4_f32: A[i = lane&15][kk = lane>>4], B[kk][j = lane&15];
//   kk indexes 4 consecutive rows of M.  Tile (ti,tj) accumulates
//   D[i][j] += sum_rows M[row][f(ti,i)] * M[row][f(tj,j)].
//   Only upper tiles (ti <= tj) are computed; they are dealt round-robin to the 4
//   waves of a block, every wave streams the block's row slab (L1-served for
//   waves 1..3).  Each block writes one partial [KP x KP] slab; a second kernel
//   sums the slabs in block order (fixed order => bit-reproducible), adds reg*I,
//   un-permutes and mirrors, so the result is exactly symmetric.
//
// Roofline: HBM-bound, algorithmic bytes = n*k*4 (one read of M).
#include <stdlib.h>

#include "als_plan.h"
#include "common.h"

namespace lk {

constexpr int GRAM_WAVES = 4;
constexpr int GRAM_SEG = 256;  // row groups (of 4 rows) per f32 MFMA chain, see gram_wave

__host__ __device__ constexpr int gram_tiles(int NT) { return NT * (NT + 1) / 2; }

// e-th upper tile, column-major packed: e = tj*(tj+1)/2 + ti
__host__ __device__ constexpr int tile_tj(int e)
{
    int tj = 0;
    while ((tj + 1) * (tj + 2) / 2 <= e) ++tj;
    return tj;
}
__host__ __device__ constexpr int tile_ti(int e) { return e - tile_tj(e) * (tile_tj(e) + 1) / 2; }

template <int NT>
struct QVec {
    float v[NT];
};

template <int NT>
__device__ __forceinline__ QVec<NT> load_qvec(const float *p)
{
    QVec<NT> q;
    if constexpr (NT == 1) {
        q.v[0] = *p;
    } else if constexpr (NT == 2) {
        f32x2 t = *reinterpret_cast<const f32x2 *>(p);
        q.v[0] = t.x;
        q.v[1] = t.y;
    } else {
#pragma unroll
        for (int c = 0; c < NT / 4; ++c) {
            f32x4 t = *reinterpret_cast<const f32x4 *>(p + 4 * c);
            q.v[4 * c + 0] = t.x;
            q.v[4 * c + 1] = t.y;
            q.v[4 * c + 2] = t.z;
            q.v[4 * c + 3] = t.w;
        }
    }
    return q;
}

template <int NT, int W>
__device__ __forceinline__ void gram_wave(const float *__restrict__ m, int64_t row_beg,
                                          int64_t row_end, int ld, float *__restrict__ slab)
{
    constexpr int NTILES = gram_tiles(NT);
    constexpr int NLOC = (NTILES - W + GRAM_WAVES - 1) / GRAM_WAVES;  // tiles of this wave
    constexpr int KP = NT * 16;
    const int lane = lane_id();
    const int sub = lane & 15, slot = lane >> 4;

    // Two-level sum: `acc` is the f32 MFMA chain of at most GRAM_SEG row groups (4 rows each);
    // it is folded into `tot` and restarted every GRAM_SEG groups.  A single chain over the
    // 78 000 rows a block used to take at n = 10^7 is ~sqrt(19 500) roundings deep (relative
    // error ~8e-6 -- times cond(A) ~ 250 that alone put ALS rows 1e-4 from float64 at cfg5);
    // chains of 256 steps + a short sum of chains stay at ~20 roundings.  Fixed order, so still
    // bit-reproducible.
    f32x4 acc[NLOC > 0 ? NLOC : 1], tot[NLOC > 0 ? NLOC : 1];
#pragma unroll
    for (int l = 0; l < NLOC; ++l) {
        acc[l] = f32x4{0.f, 0.f, 0.f, 0.f};
        tot[l] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

#ifndef LK_GRAM_PF_WIDE
#define LK_GRAM_PF_WIDE 2
#endif
    constexpr int PF = (NT >= 8) ? LK_GRAM_PF_WIDE : 4;  // groups in flight
    QVec<NT> qn[PF];
    const int64_t ngroups = (row_end - row_beg + 3) >> 2;
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        int64_t r = row_beg + (int64_t)p * 4 + slot;
        if (r < row_end)
            qn[p] = load_qvec<NT>(m + r * ld + sub * NT);
        else {
#pragma unroll
            for (int t = 0; t < NT; ++t) qn[p].v[t] = 0.f;
        }
    }
    for (int64_t g0 = 0; g0 < ngroups; g0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            QVec<NT> q = qn[p];
            int64_t rn = row_beg + (g0 + p + PF) * 4 + slot;
            if (rn < row_end)
                qn[p] = load_qvec<NT>(m + rn * ld + sub * NT);
            else {
#pragma unroll
                for (int t = 0; t < NT; ++t) qn[p].v[t] = 0.f;
            }
#pragma unroll
            for (int l = 0; l < NLOC; ++l) {
                const int e = l * GRAM_WAVES + W;
                const int ti = tile_ti(e), tj = tile_tj(e);
                acc[l] = __builtin_amdgcn_mfma_f32_16x16x4f32(q.v[ti], q.v[tj], acc[l], 0, 0, 0);
            }
        }
        if (((g0 + PF) & (GRAM_SEG - 1)) == 0) {  // wave-uniform; PF divides GRAM_SEG
#pragma unroll
            for (int l = 0; l < NLOC; ++l) {
                tot[l] += acc[l];
                acc[l] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
#pragma unroll
    for (int l = 0; l < NLOC; ++l) acc[l] += tot[l];
    // D[i = slot*4 + r][j = sub]  ->  slab[(ti*16 + i) * KP + tj*16 + j]
#pragma unroll
    for (int l = 0; l < NLOC; ++l) {
        const int e = l * GRAM_WAVES + W;
        const int ti = tile_ti(e), tj = tile_tj(e);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            slab[(ti * 16 + slot * 4 + r) * KP + tj * 16 + sub] = acc[l][r];
    }
}

template <int NT>
__global__ __launch_bounds__(256) void gramian_partial_kernel(const float *__restrict__ m,
                                                              int64_t n, int ld,
                                                              int64_t rows_per_block,
                                                              float *__restrict__ ws)
{
    constexpr int KP = NT * 16;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int64_t row_beg = (int64_t)blockIdx.x * rows_per_block;
    int64_t row_end = row_beg + rows_per_block;
    if (row_end > n) row_end = n;
    if (row_beg > n) row_beg = n;
    float *slab = ws + (size_t)blockIdx.x * KP * KP;
    switch (wave) {
        case 0: gram_wave<NT, 0>(m, row_beg, row_end, ld, slab); break;
        case 1: gram_wave<NT, 1>(m, row_beg, row_end, ld, slab); break;
        case 2: gram_wave<NT, 2>(m, row_beg, row_end, ld, slab); break;
        default: gram_wave<NT, 3>(m, row_beg, row_end, ld, slab); break;
    }
}

// ---- k = 256: the row stream staged through LDS ------------------------------------------------
//
// gramian_partial_kernel<16> holds 136 + 136 accumulator registers per wave (one wave per SIMD) and
// can keep only two 4-row groups of operands in flight next to them: 34 MFMAs x 32 cycles per group
// cover 2.2 k cycles, less than a loaded HBM round trip -- the matrix cores waited (10.7 ms per
// cfg5 epoch for 4.4 ms of MFMA work).  Here the rows arrive by `global_load_lds_dwordx4` (a wave's
// instruction moves one 1 KiB row, no register involved) into a ring of GRAM_DMA_STAGES stages of
// 16 rows: five stages = 80 KiB per CU are in flight while one is multiplied, one barrier per stage
// (4 groups = 136 MFMAs per wave) publishes everyone's rows.  The bank swizzle sits on the global
// side (position 4 s + i of a row holds its 16-byte chunk 4 s + (i ^ (s >> 2))), as in
// als_blk_chunk_dma_kernel.  Same tiles, same row order, same chain breaks: the slabs are
// bit-identical to gramian_partial_kernel<16>'s.
#ifndef LK_GRAM_DMA_STAGES
#define LK_GRAM_DMA_STAGES 4
#endif
constexpr int GRAM_DMA_STAGES = LK_GRAM_DMA_STAGES;
constexpr size_t GRAM_DMA_LDS_BYTES = (size_t)GRAM_DMA_STAGES * 16 * 1024;

template <int N>
__device__ __forceinline__ void gram_wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int W>
__device__ __forceinline__ void gram_wave_dma(const float *__restrict__ m, int64_t row_beg,
                                              int64_t row_end, float *__restrict__ slab,
                                              float *ring)
{
    constexpr int NT = 16, KP = 256, S = GRAM_DMA_STAGES;
    constexpr int NTILES = gram_tiles(NT);
    constexpr int NLOC = (NTILES - W + GRAM_WAVES - 1) / GRAM_WAVES;
    static_assert(S >= 3 && (S - 2) * 4 <= 60, "ring depth");
    const int lane = lane_id();
    const int sub = lane & 15, slot = lane >> 4;
    // (the second level of gram_wave's two-level sum -- `tot` -- lives in the block's slab in
    // global memory here, not in 136 more registers: read-modify-written every GRAM_SEG groups,
    // same float32 additions in the same order, and the kernel fits two workgroups per CU)
    f32x4 acc[NLOC];
#pragma unroll
    for (int l = 0; l < NLOC; ++l) acc[l] = f32x4{0.f, 0.f, 0.f, 0.f};
    bool flushed = false;
    const int64_t nrows = row_end - row_beg;
    const int n_stage = (int)((nrows + 15) >> 4);
    const unsigned ring_lds = (unsigned)(uintptr_t) reinterpret_cast<void *>(ring);
    // global side of this lane's 16 bytes of a row; LDS side of the operand chunks of lane
    // (entry slot, element sub): chunk c of the lane's 16 features = row chunk 4 sub + c
    const int ds_ = lane >> 2, di_ = lane & 3;
    const int src_off = 4 * (4 * ds_ + (di_ ^ (ds_ >> 2)));  // floats
    const float *rd = ring + slot * KP + 16 * sub;
    int xo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) xo[c] = 4 * (c ^ (sub >> 2));
    const int64_t last_row = row_end - 1;

    // wave W brings rows 4 g + W of the stage (g = 0..3)
    auto issue = [&](int k) {
        const int sl = k % S;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            int64_t r = row_beg + 16 * (int64_t)k + 4 * g + W;
            r = r < row_end ? r : last_row;  // (masked at use)
            const float *src = m + r * KP + src_off;
            const unsigned dst = ring_lds + (unsigned)(sl * 16 + 4 * g + W) * 1024u;
            unsigned keep;
            asm volatile(
                "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                : "=&s"(keep)
                : "v"(src), "s"(dst)
                : "memory");
        }
    };
#pragma unroll
    for (int k = 0; k < S - 1; ++k)
        if (k < n_stage) issue(k);

    for (int k = 0; k < n_stage; ++k) {
        // this wave's rows of stage k have landed when at most the rows of the stages issued
        // after it (up to S - 2 of them, 4 loads each) are still in flight
        const int ahead = n_stage - 1 - k < S - 2 ? n_stage - 1 - k : S - 2;
        switch (ahead) {
            case 0: gram_wait_vm<0>(); break;
            case 1: gram_wait_vm<4>(); break;
            case 2: gram_wait_vm<8>(); break;
            case 3: gram_wait_vm<12>(); break;
            default: gram_wait_vm<(S - 2) * 4>(); break;
        }
        asm volatile("s_barrier" ::: "memory");  // everyone's rows; everyone done with stage k - 1
        if (k + S - 1 < n_stage) issue(k + S - 1);
        const float *sp = rd + (k % S) * (16 * KP);
        const bool full = 16 * (int64_t)(k + 1) <= nrows;  // workgroup-uniform
        // operands of group g + 1 are read while the 34 MFMAs of group g run
        auto fetch = [&](float (&q)[NT], int g) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 t = *reinterpret_cast<const f32x4 *>(sp + g * (4 * KP) + xo[c]);
                q[4 * c + 0] = t.x;
                q[4 * c + 1] = t.y;
                q[4 * c + 2] = t.z;
                q[4 * c + 3] = t.w;
            }
            if (!full) {
                const bool live = 16 * (int64_t)k + 4 * g + slot < nrows;
#pragma unroll
                for (int t = 0; t < NT; ++t) q[t] = live ? q[t] : 0.f;
            }
        };
        float qc[NT], qn[NT];
        fetch(qc, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g + 1 < 4) fetch(qn, g + 1);
#pragma unroll
            for (int l = 0; l < NLOC; ++l) {
                const int e = l * GRAM_WAVES + W;
                const int ti = tile_ti(e), tj = tile_tj(e);
                acc[l] = __builtin_amdgcn_mfma_f32_16x16x4f32(qc[ti], qc[tj], acc[l], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) qc[t] = qn[t];
            __builtin_amdgcn_sched_barrier(0);
        }
        if ((((k + 1) * 4) & (GRAM_SEG - 1)) == 0) {  // every GRAM_SEG groups, as gram_wave
            // (the tile addresses are derived from a base the compiler cannot see through: hoisted
            // out of the row loop, 136 of them would be live across it)
            float *sb = slab + (slot * 4) * KP + sub;
            asm volatile("" : "+v"(sb));
#pragma unroll
            for (int l = 0; l < NLOC; ++l) {
                const int e = l * GRAM_WAVES + W;
                const int ti = tile_ti(e), tj = tile_tj(e);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float *p = sb + (ti * 16 + r) * KP + tj * 16;
                    *p = (flushed ? *p : 0.f) + acc[l][r];  // tot += acc
                }
                acc[l] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            flushed = true;
        }
    }
    float *sb = slab + (slot * 4) * KP + sub;
    asm volatile("" : "+v"(sb));
#pragma unroll
    for (int l = 0; l < NLOC; ++l) {
        const int e = l * GRAM_WAVES + W;
        const int ti = tile_ti(e), tj = tile_tj(e);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float *p = sb + (ti * 16 + r) * KP + tj * 16;
            *p = acc[l][r] + (flushed ? *p : 0.f);  // acc += tot
        }
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gramian_partial_dma_kernel(const float *__restrict__ m,
                                                                  int64_t n, int64_t rows_per_block,
                                                                  float *__restrict__ ws)
{
    constexpr int KP = 256;
    extern __shared__ __attribute__((aligned(1024))) float gram_dma_lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int64_t row_beg = (int64_t)blockIdx.x * rows_per_block;
    int64_t row_end = row_beg + rows_per_block;
    if (row_end > n) row_end = n;
    if (row_beg > n) row_beg = n;
    float *slab = ws + (size_t)blockIdx.x * KP * KP;
    switch (wave) {
        case 0: gram_wave_dma<0>(m, row_beg, row_end, slab, gram_dma_lds); break;
        case 1: gram_wave_dma<1>(m, row_beg, row_end, slab, gram_dma_lds); break;
        case 2: gram_wave_dma<2>(m, row_beg, row_end, slab, gram_dma_lds); break;
        default: gram_wave_dma<3>(m, row_beg, row_end, slab, gram_dma_lds); break;
    }
}

// LK_GRAM_DMA=0: the register-staged kernel at k = 256 too (A/B timing, tests)
static bool gram_dma_enabled()
{
    const char *e = getenv("LK_GRAM_DMA");
    return !(e && e[0] == '0');
}

// Sum the slabs (fixed order => bit-reproducible), add reg*I, un-permute, mirror.
// One workgroup per 64 consecutive primed elements (one coalesced 256-byte segment of
// every slab); wave w sums slabs w, w+4, ...; the four partial sums are combined in
// wave order.
template <int NT>
__global__ __launch_bounds__(256) void gramian_finish_kernel(const float *__restrict__ ws,
                                                             int nblocks, int k, float reg,
                                                             float *__restrict__ out, int ld_out)
{
    constexpr int KP = NT * 16;
    __shared__ double part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + lane;
    const int pr = idx / KP, pc = idx % KP;
    const int ti = pr >> 4, tj = pc >> 4;
    double sd = 0.0;
    if (ti <= tj) {  // lower tiles are never written by the partial kernel
        // the slabs are summed in float64 (up to 512 of them: a float32 sum would add another
        // ~sqrt(512 / 8) roundings on top of the chains'); one rounding to float32 at the end
        // eight loads in flight per lane: the kernel is 64 ... 1024 workgroups of dependent
        // 256-byte reads, i.e. latency (k = 64: 30 us with two in flight -- 1 % of a cfg2 epoch)
        const float *src = ws + idx;
        double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        int b = wave;
        for (; b + 28 < nblocks; b += 32) {
            float x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = src[(size_t)(b + 4 * u) * KP * KP];
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += (double)x[u];
        }
        for (int u = 0; b < nblocks; b += 4, ++u) s[u] += (double)src[(size_t)b * KP * KP];
        sd = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    }
    part[wave][lane] = sd;
    __syncthreads();
    if (wave != 0 || ti > tj) return;
    float s = (float)(((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane]);
    const int fr = (pr & 15) * NT + ti, fc = (pc & 15) * NT + tj;
    if (fr >= k || fc >= k) return;
    // within a diagonal tile both (fr,fc) and (fc,fr) are present and bitwise equal
    // (a*b == b*a, same accumulation order), so taking fr <= fc is enough.
    if (ti == tj && fr > fc) return;
    if (fr == fc) s += reg;
    out[fr * ld_out + fc] = s;
    out[fc * ld_out + fr] = s;
}

static int gram_blocks(int64_t n, int KP)
{
    // enough blocks to fill 256 CUs twice over (the workspace holds 512 slabs of KP*KP floats)
    (void)KP;
    int64_t maxb = 512;
    int64_t b = (n + 255) / 256;  // at least 256 rows (64 groups) per block
    if (b > maxb) b = maxb;
    if (b < 1) b = 1;
    return (int)b;
}

template <int NT>
static int launch_gramian(const float *m, int64_t n, int k, int ld, float reg, float *out,
                          int ld_out, float *ws, hipStream_t st)
{
    constexpr int KP = NT * 16;
    int nb = gram_blocks(n, KP);
    // rows per block: a multiple of 4 * PF groups so that every block's chains break at the
    // same places whatever n is
    int64_t rpb = ((n + nb - 1) / nb + 15) / 16 * 16;
    if (rpb < 16) rpb = 16;
    bool dma = false;
    if constexpr (NT == 16) dma = gram_dma_enabled() && n >= 16;
    if (dma) {
        static PerDeviceOnce attr_once;
        bool &attr_set = attr_once.flag();
        if (!attr_set) {
            LK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&gramian_partial_dma_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)GRAM_DMA_LDS_BYTES));
            attr_set = true;
        }
        hipLaunchKernelGGL(gramian_partial_dma_kernel, dim3(nb), dim3(256), GRAM_DMA_LDS_BYTES, st, m,
                           n, rpb, ws);
    } else {
        hipLaunchKernelGGL(gramian_partial_kernel<NT>, dim3(nb), dim3(256), 0, st, m, n, ld, rpb, ws);
    }
    hipLaunchKernelGGL(gramian_finish_kernel<NT>, dim3(KP * KP / 64), dim3(256), 0, st, ws, nb, k,
                       reg, out, ld_out);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

}  // namespace lk

extern "C" size_t lk_gramian_workspace_bytes(int32_t k)
{
    int KP = lk_padded_dim(k);
    if (KP == 0) return 0;
    if (KP > 256) return lk::gramian_big_workspace_bytes(KP);  // als_big.hip
    return (size_t)512 * KP * KP * sizeof(float);
}

extern "C" int lk_gramian(const float *d_m, int64_t n, int32_t k, int32_t ld, float reg,
                          float *d_out, int32_t ld_out, void *d_ws, void *stream)
{
    int KP = lk_padded_dim(k);
    LK_REQUIRE(KP > 0, "lk_gramian: unsupported k=%d", k);
    LK_REQUIRE(ld == KP, "lk_gramian: ld=%d must equal lk_padded_dim(k)=%d", ld, KP);
    LK_REQUIRE(ld_out >= k, "lk_gramian: ld_out=%d < k=%d", ld_out, k);
    LK_REQUIRE(d_m && d_out && d_ws, "lk_gramian: null pointer");
    LK_REQUIRE(n >= 0, "lk_gramian: negative n");
    hipStream_t st = lk::as_stream(stream);
    float *ws = static_cast<float *>(d_ws);
    switch (KP) {
        case 16: return lk::launch_gramian<1>(d_m, n, k, ld, reg, d_out, ld_out, ws, st);
        case 32: return lk::launch_gramian<2>(d_m, n, k, ld, reg, d_out, ld_out, ws, st);
        case 64: return lk::launch_gramian<4>(d_m, n, k, ld, reg, d_out, ld_out, ws, st);
        case 128: return lk::launch_gramian<8>(d_m, n, k, ld, reg, d_out, ld_out, ws, st);
        case 256: return lk::launch_gramian<16>(d_m, n, k, ld, reg, d_out, ld_out, ws, st);
    }
    if (KP > 256) return lk::gramian_big(d_m, n, k, KP, reg, d_out, ld_out, ws, st);  // als_big.hip
    return LK_E_INVALID;
}
