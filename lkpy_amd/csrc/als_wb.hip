// als_wb.hip -- implicit-ALS row solve for SHORT rows at large k (padded k = 128 / 256) through
// the Woodbury identity, gfx950.
//
// `train_row_solve` (src/accel/als/implicit.rs:87-125) solves, for every row,
//     A x = y,   A = OtOr + sum_j v_j q_j q_j^T,   y = sum_j (v_j + 1) q_j
// with a dense k x k Cholesky (sposv, src/accel/als/solve.rs:65-107): k^3/3 flops per row
// whatever the row's length n.  For n << k (cfg5: mean 10 entries, k = 256 -- 94 % of the user
// rows have n <= 16) A is a rank-n update of the SAME matrix G = OtOr for every row of the
// half-epoch, and with G^-1 known the solve collapses to an n x n system:
//     M' = diag(sqrt v) M (rows sqrt(v_j) q_j^T),   A = G + M'^T M'
//     A^-1 = G^-1 - G^-1 M'^T (I + M' G^-1 M'^T)^-1 M' G^-1
// With z_j = G^-1 q_j (rows of Z = Q G^-1, ONE plain GEMM per half-epoch), w_j = v_j + 1 and
// S0 = [q_i . z_j] (n x n):
//     S = I + diag(sqrt v) S0 diag(sqrt v),   S u = sqrt(v) o (S0 w),   x = sum_j (w_j - sqrt(v_j) u_j) z_j
// -- the same x in exact arithmetic (a direct method, no iteration, no tolerance), for
// O(n^2 k) instead of k^3/3 flops: ~200x fewer at n = 10, k = 256.  G^-1 is computed once per
// half-epoch in float64 by the host side (`ALSPlan.half_epoch`), so the error of this path is
// dominated by the f32 dot products, like the reference's.
//
// Kernel: ONE WAVE PER ROW, rows with n <= 16 (plan order is longest-first: they are a suffix
// of it, empty rows included).  Lane (s, c) = (feature quarter s, entry slot c) loads the
// quarter s of q_c and of z_c (KP/4 contiguous floats each: the 16 lanes of a quarter read 16
// rows, a row is read by 4 lanes as 4 contiguous pieces).  S0 is ONE 16 x 16 MFMA tile:
// v_mfma_f32_16x16x4_f32 with A = the lane's q values, B = its z values, summed over the
// quarter's features (the MFMA's k index is the quarter).  The 16 x 16 system is solved in lanes
// 0..15 (lane = row, v_readlane multipliers), and x = sum_j g_j z_j is a 16-lane DPP butterfly
// over the registers that already hold z.  Entry slots >= n carry zero weights.
//
// Bound: gather bandwidth (2 n rows of 4 KP bytes per solved row) -- cfg5 user half: 22 M
// entries x 2 KiB = 45 GB.
#include <type_traits>

#include "als_plan.h"
#include "common.h"

namespace lk {
namespace wb {

template <int CTRL>
__device__ __forceinline__ float dpp_add(float x)
{
    const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false);
    return x + __builtin_bit_cast(float, y);
}
// sum over the first 4 / 8 / 16 lanes of a row group (DEPTH = 2 / 3 / 4 butterfly steps): the
// result is in lane 0 of the group (DEPTH = 4: in every lane).  Entry slots >= n carry zeros,
// so a row with n <= 4 (82 % of the cfg5 user rows) needs two steps, not four.
template <int DEPTH>
__device__ __forceinline__ float row_sum(float x)
{
    x = dpp_add<0xB1>(x);  // quad_perm [1,0,3,2]
    x = dpp_add<0x4E>(x);  // quad_perm [2,3,0,1]
    if constexpr (DEPTH >= 3) x = dpp_add<0x141>(x);  // row_half_mirror
    if constexpr (DEPTH >= 4) x = dpp_add<0x140>(x);  // row_mirror
    return x;
}

constexpr int WB_LDS = 16 * 16 + 16;  // S (or L) tile + right-hand side, per wave

template <int KP, bool IS64>
__global__ __launch_bounds__(256) void als_wb_kernel(
    const typename IndPtr<IS64>::type *__restrict__ indptr, const int32_t *__restrict__ indices,
    const float *__restrict__ values, const int32_t *__restrict__ order, int64_t n_tasks,
    const float *__restrict__ other, const float *__restrict__ z, float *__restrict__ this_,
    float *__restrict__ row_delta, int *__restrict__ status)
{
    constexpr int QF = KP / 4;  // features per quarter
    constexpr int NQ = QF / 4;  // float4 per quarter
    __shared__ __attribute__((aligned(16))) float lds_all[4][WB_LDS];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const int s = lane >> 4, c = lane & 15;
    const int64_t t = (int64_t)blockIdx.x * 4 + wave;
    if (t >= n_tasks) return;
    if (status[1] != 0) return;  // Z unavailable (OtOr not positive definite): dense fallback
    const int row = order[t];
    const int64_t beg = indptr[row], end = indptr[row + 1];
    const int n = __builtin_amdgcn_readfirstlane((int)(end - beg));  // 0 .. 16, wave-uniform
    float *xrow = this_ + (int64_t)row * KP;
    float *lds = lds_all[wave];

    if (n == 0) {  // implicit.rs:98-101
        for (int f = lane; f < KP; f += 64) xrow[f] = 0.f;
        if (lane == 0) row_delta[row] = 0.f;
        return;
    }
    // entry of slot c (slots >= n re-read the last entry with zero weights)
    const int64_t e = beg + (c < n ? c : n - 1);
    const int col = indices[e];
    const float v = c < n ? values[e] : 0.f;
    const float w = c < n ? v + 1.0f : 0.f;  // `vals += 1.0` (implicit.rs:116)
    const float sv = __builtin_sqrtf(v);     // v < 0: NaN -> reported as not positive definite

    f32x4 mq[NQ], zq[NQ];
#ifndef LK_WB_MASK_LOADS
#define LK_WB_MASK_LOADS 0  // measured (cfg5, round 4): 1 = 247.0 ms/epoch, 0 = 238.4
#endif
    {
        const f32x4 *mp = reinterpret_cast<const f32x4 *>(other + (int64_t)col * KP + s * QF);
        const f32x4 *zp = reinterpret_cast<const f32x4 *>(z + (int64_t)col * KP + s * QF);
        // Entry slots >= n carry zero weights and re-read the last entry's rows (L1 hits: 32 KiB
        // through the texture path per row whatever n).  Putting ONE branch around the batch of
        // loads so that idle slots issue no request (LK_WB_MASK_LOADS=1) is SLOWER -- cfg5 247.0 vs
        // 238.4 ms/epoch: the divergent region costs more than the redundant L1 hits.
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            mq[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            zq[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (!LK_WB_MASK_LOADS || c < n) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) mq[q] = mp[q];
#pragma unroll
            for (int q = 0; q < NQ; ++q) zq[q] = zp[q];
        }
    }
    // S0[i][j] = q_i . z_j : lane (s', c') register r = S0[4 s' + r][c']
    f32x4 S0 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int el = 0; el < 4; ++el)
            S0 = __builtin_amdgcn_mfma_f32_16x16x4f32(mq[q][el], zq[q][el], S0, 0, 0, 0);

    // r0 = S0 w (row sums weighted by the column's w), S = I + diag(sv) S0 diag(sv)
    float r0[4], svi[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        // (only lane c == 0 of a group uses r0: the butterfly depth follows n)
        const float pr = S0[r] * w;
        r0[r] = n <= 4 ? row_sum<2>(pr) : (n <= 8 ? row_sum<3>(pr) : row_sum<4>(pr));
        svi[r] = __shfl(sv, 4 * s + r, 64);
    }
    f32x4 Sm;
#pragma unroll
    for (int r = 0; r < 4; ++r) Sm[r] = svi[r] * sv * S0[r] + ((4 * s + r) == c ? 1.0f : 0.f);
    // column c' of S (= row c': S is symmetric) contiguous in LDS; right-hand side behind it
    *reinterpret_cast<f32x4 *>(&lds[c * 16 + 4 * s]) = Sm;
    if (c == 0)
        *reinterpret_cast<f32x4 *>(&lds[256 + 4 * s]) =
            f32x4{svi[0] * r0[0], svi[1] * r0[1], svi[2] * r0[2], svi[3] * r0[3]};

    wave_lds_sync();
    // lane = row (lanes 0..15): right-looking Cholesky with the forward substitution folded in
    float a[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 tq = *reinterpret_cast<const f32x4 *>(&lds[(lane & 15) * 16 + 4 * q]);
        a[4 * q + 0] = tq.x;
        a[4 * q + 1] = tq.y;
        a[4 * q + 2] = tq.z;
        a[4 * q + 3] = tq.w;
    }
    float b = lds[256 + (lane & 15)];
    float minpiv = 3.0e38f, dinv = 0.f;
    // (rows / columns >= n of S are the identity: their steps are skipped, wave-uniformly)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        float lj = 0.f;
        if (j < n) {
            const float piv = bcast(a[j], j);
            minpiv = fminf(minpiv, piv);
            const float rinv = __builtin_amdgcn_rsqf(piv);
            dinv = (lane == j) ? rinv : dinv;
            lj = (lane > j && lane < 16) ? a[j] * rinv : 0.f;
            const float zj = bcast(b, j) * rinv;
            b = fmaf(-lj, zj, b);
#pragma unroll
            for (int cc = j + 1; cc < 16; ++cc)
                if (cc < n) a[cc] = fmaf(-lj, bcast(lj, cc), a[cc]);
        }
        a[j] = lj;  // row `lane` of L, strictly lower part
    }
    b *= dinv;  // z of L z = rhs
    // L^T: lane i needs column i of L -- rows of L to LDS, columns back (entries j <= i are
    // outside the strictly-lower part: cleared on the way in)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 tq;
        tq.x = (4 * q + 0 < lane) ? a[4 * q + 0] : 0.f;
        tq.y = (4 * q + 1 < lane) ? a[4 * q + 1] : 0.f;
        tq.z = (4 * q + 2 < lane) ? a[4 * q + 2] : 0.f;
        tq.w = (4 * q + 3 < lane) ? a[4 * q + 3] : 0.f;
        if (lane < 16) *reinterpret_cast<f32x4 *>(&lds[lane * 16 + 4 * q]) = tq;
    }
    wave_lds_sync();
    float lt[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) lt[j] = lds[j * 16 + (lane & 15)];  // L[j][lane], 0 for j <= lane
#pragma unroll
    for (int j = 15; j >= 1; --j)
        if (j < n) {
            const float xj = bcast(b * dinv, j);
            b = fmaf(-lt[j], xj, b);
        }
    b *= dinv;  // u' of S u' = sv o r0, lane j < 16
    // g_j = w_j - sv_j u'_j (lanes 0..15 hold entry j = lane), then to every row group
    float g = w - sv * b;
    g = __shfl(g, c, 64);

    // x = sum_j g_j z_j: this lane's quarter, summed over the 16 entry slots
    const bool bad = !(minpiv > 0.f) || !(fabsf(g) <= 3.0e38f);
    if (__any(bad) && lane == 0) atomicCAS(status, 0, row + 1);
    float d2 = 0.f;
    auto finish = [&](auto depth_c) {
        constexpr int DEPTH = decltype(depth_c)::value;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            f32x4 xs;
            xs.x = row_sum<DEPTH>(g * zq[q].x);
            xs.y = row_sum<DEPTH>(g * zq[q].y);
            xs.z = row_sum<DEPTH>(g * zq[q].z);
            xs.w = row_sum<DEPTH>(g * zq[q].w);
            if (c == 0) {
                f32x4 *dst = reinterpret_cast<f32x4 *>(xrow + s * QF + 4 * q);
                const f32x4 old = *dst;
                *dst = xs;
                const f32x4 d = xs - old;
                d2 += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
            }
        }
    };
    if (n <= 4)
        finish(std::integral_constant<int, 2>{});
    else if (n <= 8)
        finish(std::integral_constant<int, 3>{});
    else
        finish(std::integral_constant<int, 4>{});
    d2 = wave_sum(d2);
    if (lane == 0) row_delta[row] = d2;
}

// ---- FOUR rows per wave for rows with at most 4 entries (round 4) ---------------------------------
// 82 % of cfg5's 10^7 user rows have <= 4 entries.  als_wb_kernel spends a whole wave on each --
// 16 entry slots of which 2.3 are in use on average; at 9.4 M rows the kernel is bound by its
// instruction stream and the dependent latencies of a row (1.5 TB/s of traffic), not by memory.
// Here entry slot c = 4 r4 + e belongs to row r4 = c >> 2 of the wave (entry e = c & 3): the same
// loads, the same single MFMA tile -- whose four DIAGONAL 4 x 4 blocks are the four rows' S0
// (everything off those blocks is masked to zero, so S is block diagonal and one 16-step
// factorisation solves the four independent systems) -- the same DPP quad sums, a quarter of the
// waves.  Empty rows (n = 0) ride along: zero weights give x = 0; their delta is forced to 0
// (implicit.rs:98-101).
// B = entry slots per row (4 or 8), 16 / B rows per wave
template <int KP, bool IS64, int B = 4>
__global__ __launch_bounds__(256) void als_wb4_kernel(
    const typename IndPtr<IS64>::type *__restrict__ indptr, const int32_t *__restrict__ indices,
    const float *__restrict__ values, const int32_t *__restrict__ order, int64_t n_tasks,
    const float *__restrict__ other, const float *__restrict__ z, float *__restrict__ this_,
    float *__restrict__ row_delta, int *__restrict__ status)
{
    constexpr int QF = KP / 4;  // features per quarter
    constexpr int NQ = QF / 4;  // float4 per quarter
    __shared__ __attribute__((aligned(16))) float lds_all[4][WB_LDS];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const int s = lane >> 4, c = lane & 15;
    static_assert(B == 4 || B == 8, "4 or 8 entry slots per row");
    constexpr int RPW = 16 / B;          // rows per wave
    constexpr int DEPTH = B == 4 ? 2 : 3;  // butterfly steps over a row's entry slots
    const int r4 = c / B, e4 = c % B;    // row of the wave, entry of the row
    const int64_t t = ((int64_t)blockIdx.x * 4 + wave) * RPW + r4;  // this lane's row task
    if (((int64_t)blockIdx.x * 4 + wave) * RPW >= n_tasks) return;  // wave-uniform
    if (status[1] != 0) return;  // Z unavailable (OtOr not positive definite): dense fallback
    const bool have = t < n_tasks;
    const int row = have ? order[t] : 0;
    int64_t beg = 0;
    int n = 0;
    if (have) {
        beg = indptr[row];
        n = (int)(indptr[row + 1] - beg);  // 0 .. B
    }
    float *xrow = this_ + (int64_t)row * KP;
    float *lds = lds_all[wave];

    // entry e4 of this lane's row (slots >= n: the row's last entry -- or entry 0 of the matrix
    // for an empty row -- with zero weights)
    const bool live = e4 < n;
    const int64_t e = n > 0 ? beg + (live ? e4 : n - 1) : 0;
    const int col = n > 0 ? indices[e] : 0;  // (an empty row reads factor row 0 with zero weights)
    const float v = live ? values[e] : 0.f;
    const float w = live ? v + 1.0f : 0.f;  // `vals += 1.0` (implicit.rs:116)
    const float sv = __builtin_sqrtf(v);    // v < 0: NaN -> reported as not positive definite

    f32x4 mq[NQ], zq[NQ];
    {
        const f32x4 *mp = reinterpret_cast<const f32x4 *>(other + (int64_t)col * KP + s * QF);
        const f32x4 *zp = reinterpret_cast<const f32x4 *>(z + (int64_t)col * KP + s * QF);
#pragma unroll
        for (int q = 0; q < NQ; ++q) mq[q] = mp[q];
#pragma unroll
        for (int q = 0; q < NQ; ++q) zq[q] = zp[q];
    }
    // S0[i][j] = q_i . z_j : lane (s', c') register r = S0[4 s' + r][c']; only the diagonal 4 x 4
    // blocks (4 s' + r and c' in the same row of the wave, i.e. c' >> 2 == s') mean anything
    f32x4 S0 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int el = 0; el < 4; ++el)
            S0 = __builtin_amdgcn_mfma_f32_16x16x4f32(mq[q][el], zq[q][el], S0, 0, 0, 0);
    const bool inblock = r4 == (4 * s) / B;  // (rows 4 s .. 4 s + 3 lie in one block)
#pragma unroll
    for (int r = 0; r < 4; ++r) S0[r] = inblock ? S0[r] : 0.f;

    float r0[4], svi[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        r0[r] = row_sum<DEPTH>(S0[r] * w);  // over the row's B entry slots, in each of their lanes
        svi[r] = __shfl(sv, 4 * s + r, 64);
    }
    f32x4 Sm;
#pragma unroll
    for (int r = 0; r < 4; ++r) Sm[r] = svi[r] * sv * S0[r] + ((4 * s + r) == c ? 1.0f : 0.f);
    *reinterpret_cast<f32x4 *>(&lds[c * 16 + 4 * s]) = Sm;
    // right-hand side of system rows 4 s .. 4 s + 3: held by the lanes of their block's entry
    // slots in row group s; its leader writes
    if (c == B * ((4 * s) / B))
        *reinterpret_cast<f32x4 *>(&lds[256 + 4 * s]) =
            f32x4{svi[0] * r0[0], svi[1] * r0[1], svi[2] * r0[2], svi[3] * r0[3]};

    wave_lds_sync();
    // lane = row (lanes 0..15): right-looking Cholesky with the forward substitution folded in
    float a[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 tq = *reinterpret_cast<const f32x4 *>(&lds[(lane & 15) * 16 + 4 * q]);
        a[4 * q + 0] = tq.x;
        a[4 * q + 1] = tq.y;
        a[4 * q + 2] = tq.z;
        a[4 * q + 3] = tq.w;
    }
    float b = lds[256 + (lane & 15)];
    float dinv = 0.f, mypiv = 1.0f;
#ifdef LK_WB4_DEBUG
    const float dbg_rhs = b;
    float dbg_diag = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) dbg_diag = (lane == j) ? a[j] : dbg_diag;
#endif
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float piv = bcast(a[j], j);
        mypiv = (lane == j) ? piv : mypiv;
        const float rinv = __builtin_amdgcn_rsqf(piv);
        dinv = (lane == j) ? rinv : dinv;
        const float lj = (lane > j && lane < 16) ? a[j] * rinv : 0.f;
        const float zj = bcast(b, j) * rinv;
        b = fmaf(-lj, zj, b);
        // (S is block diagonal: only the columns of the same B x B block can be non-zero)
#pragma unroll
        for (int cc = j + 1; cc < (j | (B - 1)) + 1; ++cc) a[cc] = fmaf(-lj, bcast(lj, cc), a[cc]);
        a[j] = lj;  // row `lane` of L, strictly lower part
    }
    b *= dinv;  // z of L z = rhs
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 tq;
        tq.x = (4 * q + 0 < lane) ? a[4 * q + 0] : 0.f;
        tq.y = (4 * q + 1 < lane) ? a[4 * q + 1] : 0.f;
        tq.z = (4 * q + 2 < lane) ? a[4 * q + 2] : 0.f;
        tq.w = (4 * q + 3 < lane) ? a[4 * q + 3] : 0.f;
        if (lane < 16) *reinterpret_cast<f32x4 *>(&lds[lane * 16 + 4 * q]) = tq;
    }
    wave_lds_sync();
    float lt[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) lt[j] = lds[j * 16 + (lane & 15)];  // L[j][lane], 0 for j <= lane
#pragma unroll
    for (int j = 15; j >= 1; --j) {
        const float xj = bcast(b * dinv, j);
        b = fmaf(-lt[j], xj, b);
    }
    b *= dinv;  // u' of S u' = sv o r0, lane j < 16
    float g = w - sv * b;  // lanes 0..15 hold entry slot j = lane
#ifdef LK_WB4_DEBUG
    {  // dump of the wave's system into the x row of the wave's first task
        float *dbg = this_ + (int64_t)__builtin_amdgcn_readfirstlane(row) * KP;
        if (lane < 16) {
            dbg[lane] = b;
            dbg[16 + lane] = dbg_rhs;
            dbg[32 + lane] = mypiv;
            dbg[48 + lane] = sv;
            dbg[64 + lane] = w;
            dbg[80 + lane] = dinv;
            dbg[96 + lane] = dbg_diag;
        }
        return;
    }
#endif
    g = __shfl(g, c, 64);
    // not positive definite / NaN: reported with the row of the slot's block
    {
        const bool badc = lane < 16 && (!(mypiv > 0.f) || !(fabsf(w - sv * b) <= 3.0e38f));
        if (badc && have) atomicCAS(status, 0, row + 1);
    }

    // x = sum over the row's 4 entry slots of g z: quad sums, the quad leader writes its quarter
    float d2 = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        f32x4 xs;
        xs.x = row_sum<DEPTH>(g * zq[q].x);
        xs.y = row_sum<DEPTH>(g * zq[q].y);
        xs.z = row_sum<DEPTH>(g * zq[q].z);
        xs.w = row_sum<DEPTH>(g * zq[q].w);
        if (e4 == 0 && have) {
            f32x4 *dst = reinterpret_cast<f32x4 *>(xrow + s * QF + 4 * q);
            const f32x4 old = *dst;
            *dst = xs;
            const f32x4 d = xs - old;
            d2 += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
        }
    }
    // the row's delta: its four quarter leaders are the lanes (s = 0..3, c = B r4)
    float tot = 0.f;
#pragma unroll
    for (int ss = 0; ss < 4; ++ss) tot += __shfl(d2, 16 * ss + B * r4, 64);
    if (s == 0 && e4 == 0 && have) row_delta[row] = n > 0 ? tot : 0.f;  // implicit.rs:98-101
}

}  // namespace wb

// rows [t0, n_rows) of the plan order (n <= `slots` entries each, empty rows included), 16 / slots
// per wave (slots = 4 or 8)
int als_wb4_launch(const lk_als_plan *p, const void *indptr, int is64, const int32_t *indices,
                   const float *values, int64_t t0, int64_t n_rows, float *this_,
                   const float *other, const float *z, float *row_delta, int *status,
                   hipStream_t st, int slots)
{
    const int64_t n = n_rows - t0;
    if (n <= 0) return LK_OK;
    const int64_t per_wg = 4 * (16 / slots);
    const dim3 grid((unsigned)((n + per_wg - 1) / per_wg)), block(256);
#define LK_WB4(KPV, IS, BV)                                                                     \
    hipLaunchKernelGGL((wb::als_wb4_kernel<KPV, IS, BV>), grid, block, 0, st,                   \
                       static_cast<const typename IndPtr<IS>::type *>(indptr), indices, values, \
                       p->d_order + t0, n, other, z, this_, row_delta, status)
#define LK_WB4_IS(KPV, BV)     \
    do {                       \
        if (is64)              \
            LK_WB4(KPV, true, BV);  \
        else                   \
            LK_WB4(KPV, false, BV); \
    } while (0)
    if (slots != 4 && slots != 8) {
        set_error("Woodbury row solve: %d entry slots per row", slots);
        return LK_E_INVALID;
    }
    if (p->KP == 256) {
        if (slots == 4)
            LK_WB4_IS(256, 4);
        else
            LK_WB4_IS(256, 8);
    } else if (p->KP == 128) {
        if (slots == 4)
            LK_WB4_IS(128, 4);
        else
            LK_WB4_IS(128, 8);
    } else {
        set_error("Woodbury row solve: unsupported padded embedding size %d", p->KP);
        return LK_E_INVALID;
    }
#undef LK_WB4_IS
#undef LK_WB4
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

// rows [t0, n_rows) of the plan order (n <= 16 entries each) through the Woodbury kernel
int als_wb_launch(const lk_als_plan *p, const void *indptr, int is64, const int32_t *indices,
                  const float *values, int64_t t0, int64_t n_rows, float *this_,
                  const float *other, const float *z, float *row_delta, int *status,
                  hipStream_t st)
{
    const int64_t n = n_rows - t0;
    if (n <= 0) return LK_OK;
    const dim3 grid((unsigned)((n + 3) / 4)), block(256);
#define LK_WB(KPV, IS)                                                                          \
    hipLaunchKernelGGL((wb::als_wb_kernel<KPV, IS>), grid, block, 0, st,                        \
                       static_cast<const typename IndPtr<IS>::type *>(indptr), indices, values, \
                       p->d_order + t0, n, other, z, this_, row_delta, status)
    if (p->KP == 256) {
        if (is64)
            LK_WB(256, true);
        else
            LK_WB(256, false);
    } else if (p->KP == 128) {
        if (is64)
            LK_WB(128, true);
        else
            LK_WB(128, false);
    } else {
        set_error("Woodbury row solve: unsupported padded embedding size %d", p->KP);
        return LK_E_INVALID;
    }
#undef LK_WB
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

}  // namespace lk
