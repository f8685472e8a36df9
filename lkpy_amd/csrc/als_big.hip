// als_big.hip -- exact (Cholesky) ALS row solve for embedding sizes ABOVE 256 (padded to a
// multiple of 64, up to 1024), gfx950.
//
// `POSV::solve` (src/accel/als/solve.rs:65-107) hands any k to LAPACK sposv; the register-resident
// solvers of als_chol.hip (k <= 64) and als_blk.hip (k = 128 / 256) stop at 256 -- the packed
// factor of a 512 x 512 matrix is more than a CU's registers and LDS together.  This file keeps
// the SAME blocked right-looking algorithm as als_blk.hip (16-column panels, lane = panel row
// v_readlane chain with the forward substitution riding along, MFMA trailing updates,
// back substitution block row by block row) but the 16 x 16 tiles of the normal matrix live in a
// per-row scratch in HBM / L2 instead of accumulator registers, and every tile loop is a runtime
// loop.  Per batch of rows (the scratch bounds the batch):
//
//   als_big_gram_kernel    grid = rows x tile rows: A' = OtOr + sum_j v_j q_j q_j^T (upper tiles),
//                          v_mfma_f32_16x16x4_f32, a wave holds up to 16 tiles of its tile row;
//                          explicit model: A' = sum q q^T + reg n I  (explicit.rs:103-107)
//   als_big_solve_kernel   grid = rows: y = sum_j (v_j + 1) q_j, blocked Cholesky in place on the
//                          scratch tiles, x = A'^-1 y, this[row] <- x, delta
//
// Feature order: natural (tile t = features 16 t .. 16 t + 15); no Woodbury path, no chunk slabs
// (a long row is spread over NT workgroups by the Gram kernel's grid).  Bound: the trailing
// updates stream the scratch tiles, ~NT^3 / 6 x 2 KiB per row (11 MB at k = 512): L2 / HBM
// bandwidth, not MFMA.  Written for coverage of the reference's "any k"; BASELINE's largest
// configuration is k = 256.
#include <type_traits>

#include "als_plan.h"
#include "common.h"

namespace lk {
namespace big {

constexpr int MAXT = 16;  // tiles of one tile row a wave holds in the Gram kernel: NT <= 64

__host__ __device__ inline int64_t n_pairs(int NT) { return (int64_t)NT * (NT + 1) / 2; }
// upper tile (ti <= tj) -> index, row-major by ti
__host__ __device__ inline int64_t pair_index(int NT, int ti, int tj)
{
    return (int64_t)ti * NT - (int64_t)ti * (ti - 1) / 2 + (tj - ti);
}

// OtOr [k x k] -> padded [KP x KP], identity on the pad diagonal (otor null: explicit model)
__global__ void big_prep_otor_kernel(const float *__restrict__ otor, int ld_otor, int k, int KP,
                                     float *__restrict__ otor_p)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)KP * KP) return;
    const int r = (int)(idx / KP), c = (int)(idx % KP);
    float v;
    if (r < k && c < k)
        v = otor ? otor[(int64_t)r * ld_otor + c] : 0.f;
    else
        v = (r == c) ? 1.0f : 0.0f;
    otor_p[idx] = v;
}

// ---- normal matrix: task = (row of the batch, tile row ti) -------------------------------------
// Tile (ti, tj) in accumulator layout: lane (sub = lane & 15, slot = lane >> 4), register r holds
// A'[16 ti + 4 slot + r][16 tj + sub]; stored as one float4 per lane.
template <bool IS64, bool EXPL>
__global__ __launch_bounds__(256) void als_big_gram_kernel(
    const typename IndPtr<IS64>::type *__restrict__ indptr, const int32_t *__restrict__ indices,
    const float *__restrict__ values, const int32_t *__restrict__ order, int64_t t0,
    int64_t n_batch, const float *__restrict__ other, int KP, int NT, int k, float reg,
    const float *__restrict__ otor_p, float *__restrict__ tiles)
{
    const int64_t rb = blockIdx.x / NT;
    const int ti = (int)(blockIdx.x % NT);
    if (rb >= n_batch) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const int sub = lane & 15, slot = lane >> 4;
    const int row = order[t0 + rb];
    const int64_t beg = indptr[row], end = indptr[row + 1];
    if (end == beg) return;  // empty row: the solve kernel writes zeros
    // this wave's tiles: tj = ti + wave, ti + wave + 4, ...
    const int nt_w = (NT - ti - wave + 3) / 4;  // <= MAXT
    if (nt_w <= 0) return;
    // The reference's summation order, on EVERY row (round 5): `mtl.dot(&o_picked)`
    // (src/accel/als/implicit.rs:112) is matrixmultiply's sgemm -- the row's entries in blocks of
    // KC = 256, one fma chain per block starting from zero, the block sums added one after the
    // other -- and `otor + &mtm` adds OtOr last.  `acc` is the running block, `tot` the sum of the
    // finished blocks; one unbroken chain over a 10^5-entry row would stagnate differently.
    f32x4 acc[MAXT], tot[MAXT];
#pragma unroll
    for (int q = 0; q < MAXT; ++q) {
        acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        tot[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int64_t last = end - 1;
    int in_block = 0;  // entries of the running block so far
    for (int64_t e0 = beg; e0 < end; e0 += 4) {
        const int64_t e = e0 + slot <= last ? e0 + slot : last;
        const bool live = e0 + slot <= last;
        const int col = indices[e];
        const float v = values[e];
        const float *qrow = other + (int64_t)col * KP;
        const float qa = qrow[16 * ti + sub];
        // `mtl = mt * vals` rounded first (implicit.rs:110-111); explicit: plain M^T M
        const float a = live ? (EXPL ? qa : qa * v) : 0.f;
#pragma unroll
        for (int q = 0; q < MAXT; ++q) {
            if (q < nt_w) {  // wave-uniform
                const float qb = qrow[16 * (ti + wave + 4 * q) + sub];
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, live ? qb : 0.f, acc[q], 0, 0, 0);
            }
        }
        in_block += 4;
        if (in_block == 256) {  // (wave-uniform) a finished block joins the total, in order
#pragma unroll
            for (int q = 0; q < MAXT; ++q) {
                tot[q] += acc[q];
                acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            in_block = 0;
        }
    }
#pragma unroll
    for (int q = 0; q < MAXT; ++q) {
        if (in_block > 0) tot[q] += acc[q];  // the last, partial block
        if (q < nt_w) {  // a = otor + mtm
            const int tj = ti + wave + 4 * q;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[q][r] = otor_p[(int64_t)(16 * ti + 4 * slot + r) * KP + 16 * tj + sub] + tot[q][r];
        }
    }
    const float dg = EXPL ? reg * (float)(end - beg) : 0.f;  // explicit.rs:104-107
#pragma unroll
    for (int q = 0; q < MAXT; ++q) {
        if (q < nt_w) {
            const int tj = ti + wave + 4 * q;
            f32x4 o = acc[q];
            if (EXPL && tj == ti) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * slot + r == sub && 16 * ti + sub < k) o[r] += dg;
            }
            *reinterpret_cast<f32x4 *>(tiles + ((size_t)rb * n_pairs(NT) + pair_index(NT, ti, tj)) * 256 +
                                       lane * 4) = o;
        }
    }
}

// LDS layout of the solve kernel (floats): panel P[4][KP][4] (k-group-major, as als_blk.hip),
// then y, z, x, rinv [KP] each, zb[16], sp[4][16], ld[16][16] (diagonal block), red[8]
__host__ __device__ inline size_t solve_lds_floats(int KP)
{
    return (size_t)16 * KP + 4 * (size_t)KP + 16 + 64 + 256 + 8;
}

template <bool IS64, bool EXPL>
__global__ __launch_bounds__(256) void als_big_solve_kernel(
    const typename IndPtr<IS64>::type *__restrict__ indptr, const int32_t *__restrict__ indices,
    const float *__restrict__ values, const int32_t *__restrict__ order, int64_t t0,
    int64_t n_batch, const float *__restrict__ other, float *__restrict__ this_, int KP, int NT,
    int k, float *__restrict__ tiles, float *__restrict__ ldiag, float *__restrict__ row_delta,
    int *__restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *P = lds;
    float *Y = lds + (size_t)16 * KP;
    float *Z = Y + KP;
    float *X = Z + KP;
    float *RINV = X + KP;
    float *ZB = RINV + KP;
    float *SP = ZB + 16;
    float *LD = SP + 64;
    float *RED = LD + 256;
    const int64_t rb = blockIdx.x;
    if (rb >= n_batch) return;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = lane_id();
    const int sub = lane & 15, slot = lane >> 4;
    const int row = order[t0 + rb];
    const int64_t beg = indptr[row], end = indptr[row + 1];
    float *xrow = this_ + (int64_t)row * KP;
    if (end == beg) {  // implicit.rs:98-101
        for (int f = tid; f < KP; f += 256) xrow[f] = 0.f;
        if (tid == 0) row_delta[row] = 0.f;
        return;
    }
    float *T = tiles + (size_t)rb * n_pairs(NT) * 256;
    float *LDG = ldiag + (size_t)rb * NT * 256;
    const int PSUB = KP * 4;  // one k-group sub-panel: [row][4]

    // ---- right-hand side: y = sum_j (v_j + 1) q_j  (explicit: v_j q_j), thread = feature(s) ------
    // (the reference's order: ONE sequential float32 chain per feature, product and sum rounded
    // separately -- `mt.dot(&vals)` on a strided view, implicit.rs:116-117; see als_rhs.hip)
    for (int f = tid; f < KP; f += 256) {
        float y = 0.f;
        for (int64_t e = beg; e < end; ++e) {
            const float v = values[e];
            float prod = other[(int64_t)indices[e] * KP + f] * (EXPL ? v : v + 1.0f);
            asm volatile("" : "+v"(prod));  // keep hipcc from fusing the pair into v_fmac
            y = y + prod;
        }
        Y[f] = y;
    }
    __syncthreads();

    float minpiv = 3.0e38f;
    for (int b = 0; b < NT; ++b) {
        const int R = KP - 16 * b;  // panel rows: block row b and everything below
        // (1) publish block row b (= block column b of L^T, by symmetry) as panel rows:
        // tile (b, tj), lane (sub, slot): A'[16 b + 4 slot + r][16 tj + sub] = panel row
        // 16 (tj - b) + sub, columns 4 slot .. 4 slot + 3
        for (int tj = b + wave; tj < NT; tj += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(T + pair_index(NT, b, tj) * 256 + lane * 4);
            *reinterpret_cast<f32x4 *>(&P[slot * PSUB + (16 * (tj - b) + sub) * 4]) = v;
        }
        __syncthreads();
        // (2) lane = panel row, 256 rows at a time; every wave with rows factors the diagonal
        // block again (lane & 15 = its row) for the v_readlane multipliers.  Thread 0 of the first
        // group carries the right-hand side instead of a matrix row.
        for (int rg = 0; rg * 256 < R; ++rg) {
            if (rg * 256 + wave * 64 >= R) continue;  // wave-uniform: no rows for this wave
            const int p = rg * 256 + tid;
            const int prow = p < R ? p : R - 1;
            float a[16], d[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 t = *reinterpret_cast<const f32x4 *>(&P[g * PSUB + prow * 4]);
                a[4 * g + 0] = t.x;
                a[4 * g + 1] = t.y;
                a[4 * g + 2] = t.z;
                a[4 * g + 3] = t.w;
                const f32x4 u = *reinterpret_cast<const f32x4 *>(&P[g * PSUB + sub * 4]);
                d[4 * g + 0] = u.x;
                d[4 * g + 1] = u.y;
                d[4 * g + 2] = u.z;
                d[4 * g + 3] = u.w;
            }
            if (p == 0) {
#pragma unroll
                for (int c = 0; c < 16; ++c) a[c] = Y[16 * b + c];
            }
            float myrinv = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float piv = bcast(d[j], j);
                minpiv = fminf(minpiv, piv);
                const float rinv = __builtin_amdgcn_rsqf(piv);
                myrinv = (sub == j) ? rinv : myrinv;
                d[j] *= rinv;
                a[j] *= rinv;
#pragma unroll
                for (int c = j + 1; c < 16; ++c) {
                    const float m = bcast(d[j], c);  // L[c][j]
                    d[c] = fmaf(-d[j], m, d[c]);
                    a[c] = fmaf(-a[j], m, a[c]);
                }
            }
            // L panel rows back in place; diagonal block, 1 / L_jj and z_b to their homes
            if (p >= 16 && p < R) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4 *>(&P[g * PSUB + p * 4]) =
                        f32x4{a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
            }
            if (rg == 0 && wave == 0 && lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c) LDG[b * 256 + lane * 16 + c] = c < lane ? d[c] : 0.f;
                RINV[16 * b + lane] = myrinv;
                if (lane == 0) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        ZB[c] = a[c];
                        Z[16 * b + c] = a[c];
                    }
                }
            }
        }
        __syncthreads();
        if (b + 1 < NT) {
            // (3) the finished panel L(:, b) back to the scratch (the back substitution reads it)
            for (int tj = b + 1 + wave; tj < NT; tj += 4) {
                const f32x4 v =
                    *reinterpret_cast<const f32x4 *>(&P[slot * PSUB + (16 * (tj - b) + sub) * 4]);
                *reinterpret_cast<f32x4 *>(T + pair_index(NT, b, tj) * 256 + lane * 4) = v;
            }
            // (4) forward substitution of the rows below: y_r -= L[r][b-block] . z_b
            for (int p = 16 + tid; p < R; p += 256) {
                float s = Y[16 * b + p];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 l = *reinterpret_cast<const f32x4 *>(&P[g * PSUB + p * 4]);
                    s = fmaf(-l.x, ZB[4 * g + 0], s);
                    s = fmaf(-l.y, ZB[4 * g + 1], s);
                    s = fmaf(-l.z, ZB[4 * g + 2], s);
                    s = fmaf(-l.w, ZB[4 * g + 3], s);
                }
                Y[16 * b + p] = s;
            }
            // (5) trailing update A'(ti, tj) -= L(ti, b) L(tj, b)^T for b < ti <= tj; operand of
            // MFMA step kk: lane (m = sub, kg = slot) supplies L[16 (t - b) + m][4 kg + kk]
            for (int ti = b + 1; ti < NT; ++ti) {
                const f32x4 la =
                    *reinterpret_cast<const f32x4 *>(&P[slot * PSUB + (16 * (ti - b) + sub) * 4]);
                for (int tj = ti; tj < NT; ++tj) {
                    if (((ti + tj) & 3) != wave) continue;  // wave-uniform deal
                    const f32x4 lb = *reinterpret_cast<const f32x4 *>(
                        &P[slot * PSUB + (16 * (tj - b) + sub) * 4]);
                    float *tp = T + pair_index(NT, ti, tj) * 256 + lane * 4;
                    f32x4 acc = *reinterpret_cast<const f32x4 *>(tp);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(-la[kk], lb[kk], acc, 0, 0, 0);
                    *reinterpret_cast<f32x4 *>(tp) = acc;
                }
            }
        }
        __syncthreads();  // (tiles and Y updated by other waves are read by the next step)
    }

    // ---- back substitution  L^T x = z, block row by block row from the bottom ---------------------
    for (int b = NT - 1; b >= 0; --b) {
        // s_b[c] = sum_{tj > b} sum_j L[16 tj + j][16 b + c] x[16 tj + j]; tile (b, tj) holds
        // L[16 tj + sub][16 b + 4 slot + r] in register r
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int tj = b + 1 + wave; tj < NT; tj += 4) {
            const f32x4 l = *reinterpret_cast<const f32x4 *>(T + pair_index(NT, b, tj) * 256 + lane * 4);
            const float xv = X[16 * tj + sub];
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = fmaf(l[r], xv, t[r]);
        }
#pragma unroll
        for (int m = 1; m < 16; m <<= 1)
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] += __shfl_xor(t[r], m, 64);
        if (sub == 0)
            *reinterpret_cast<f32x4 *>(&SP[wave * 16 + slot * 4]) = f32x4{t[0], t[1], t[2], t[3]};
        if (tid < 256) LD[tid] = LDG[b * 256 + tid];  // the diagonal block, [row][col] strictly lower
        __syncthreads();
        if (wave == 0) {
            const int c = lane & 15;
            float dc = Z[16 * b + c] - ((SP[c] + SP[16 + c]) + (SP[32 + c] + SP[48 + c]));
            const float ri = RINV[16 * b + c];
            float xc = 0.f;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const int j = 15 - jj;
                const float xj = bcast(dc * ri, j);
                xc = (c == j) ? xj : xc;
                dc = fmaf(-LD[j * 16 + c], xj, dc);  // L[j][c], j > c (0 otherwise)
            }
            if (lane < 16) X[16 * b + c] = xc;
        }
        __syncthreads();
    }

    // ---- output ------------------------------------------------------------------------------------
    float dd = 0.f;
    bool bad = !(minpiv > 0.f);
    for (int f = tid; f < KP; f += 256) {
        if (f < k) {
            const float x = X[f];
            const float old = xrow[f];
            xrow[f] = x;
            const float d = x - old;
            dd += d * d;
            bad = bad || !(fabsf(x) <= 3.0e38f);
        }
    }
    const float d2 = wave_sum(dd);
    if (lane == 0) RED[wave] = d2;
    if (__any(bad) && lane == 0) atomicCAS(status, 0, row + 1);
    __syncthreads();
    if (tid == 0) row_delta[row] = ((RED[0] + RED[1]) + RED[2]) + RED[3];
}

// rows of the batch the scratch holds: tiles + diagonal blocks, at most ~2 GiB
__host__ inline int64_t batch_rows(int KP, int64_t n_rows)
{
    const int NT = KP / 16;
    const size_t per = ((size_t)n_pairs(NT) + NT) * 256 * sizeof(float);
    int64_t b = (int64_t)(((size_t)2 << 30) / per);
    if (b < 64) b = 64;
    if (b > n_rows) b = n_rows;
    return b < 1 ? 1 : b;
}

}  // namespace big

size_t als_big_scratch_bytes(int KP, int64_t n_rows)
{
    const int NT = KP / 16;
    return (size_t)big::batch_rows(KP, n_rows) * ((size_t)big::n_pairs(NT) + NT) * 256 * sizeof(float);
}

template <bool IS64, bool EXPL>
static int launch_big(const lk_als_plan *p, const void *indptr, const int32_t *indices,
                      const float *values, int64_t n_rows, int k, float *this_, const float *other,
                      const float *otor, int ld_otor, char *ws, float *out_frob, hipStream_t st,
                      float reg)
{
    using IT = typename IndPtr<IS64>::type;
    const int KP = p->KP, NT = KP / 16;
    int *status = reinterpret_cast<int *>(ws + p->off_status);
    float *otor_p = reinterpret_cast<float *>(ws + p->off_otor);
    float *row_delta = reinterpret_cast<float *>(ws + p->off_delta);
    float *partial = reinterpret_cast<float *>(ws + p->off_partial);
    float *tiles = reinterpret_cast<float *>(ws + p->off_slabs);
    LK_REQUIRE(!p->ctl, "embedding sizes above 256 do not poll a task-control block");
    LK_REQUIRE(!p->d_yref && !p->ref_order, "reference-order plans stop at k = 256");
    LK_HIP_CHECK(hipMemsetAsync(status, 0, 64, st));
    hipLaunchKernelGGL(big::big_prep_otor_kernel, dim3((unsigned)(((size_t)KP * KP + 255) / 256)),
                       dim3(256), 0, st, otor, ld_otor, k, KP, otor_p);
    const int64_t B = big::batch_rows(KP, n_rows);
    float *ldiag = tiles + (size_t)B * big::n_pairs(NT) * 256;
    const size_t lds_bytes = big::solve_lds_floats(KP) * sizeof(float);
    if (lds_bytes > 64 * 1024)
        LK_HIP_CHECK(hipFuncSetAttribute(
            reinterpret_cast<const void *>(&big::als_big_solve_kernel<IS64, EXPL>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    const bool tm = p->timing && p->timing_n < lk_als_plan::TIMING_RING;
    if (tm) {
        LK_HIP_CHECK(hipEventRecord(p->ev[p->timing_n][0], st));
        LK_HIP_CHECK(hipEventRecord(p->ev[p->timing_n][1], st));
    }
    for (int64_t t0 = 0; t0 < n_rows; t0 += B) {
        const int64_t nb = n_rows - t0 < B ? n_rows - t0 : B;
        hipLaunchKernelGGL((big::als_big_gram_kernel<IS64, EXPL>), dim3((unsigned)(nb * NT)),
                           dim3(256), 0, st, static_cast<const IT *>(indptr), indices, values,
                           p->d_order, t0, nb, other, KP, NT, k, reg, otor_p, tiles);
        hipLaunchKernelGGL((big::als_big_solve_kernel<IS64, EXPL>), dim3((unsigned)nb), dim3(256),
                           lds_bytes, st, static_cast<const IT *>(indptr), indices, values,
                           p->d_order, t0, nb, other, this_, KP, NT, k, tiles, ldiag, row_delta,
                           status);
    }
    if (tm) {
        LK_HIP_CHECK(hipEventRecord(p->ev[p->timing_n][2], st));
        p->timing_n++;
    }
    int rc = launch_delta_reduce(row_delta, n_rows, partial, out_frob, st);
    if (rc != LK_OK) return rc;
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

// Exact half-epoch for padded k > 256 (dispatch target of lk_als_implicit_half_epoch /
// lk_als_explicit_half_epoch); `otor` null = explicit model.
int als_big_half_epoch(const lk_als_plan *p, const void *indptr, int is64, const int32_t *indices,
                       const float *values, int64_t n_rows, int k, float *this_,
                       const float *other, const float *otor, int ld_otor, char *ws,
                       float *out_frob, hipStream_t st, bool expl, float reg)
{
    LK_REQUIRE(p->KP > 256 && p->KP <= 1024 && p->KP % 64 == 0,
               "als_big_half_epoch: padded embedding size %d", p->KP);
    if (expl)
        return is64 ? launch_big<true, true>(p, indptr, indices, values, n_rows, k, this_, other,
                                             otor, ld_otor, ws, out_frob, st, reg)
                    : launch_big<false, true>(p, indptr, indices, values, n_rows, k, this_, other,
                                              otor, ld_otor, ws, out_frob, st, reg);
    return is64 ? launch_big<true, false>(p, indptr, indices, values, n_rows, k, this_, other,
                                          otor, ld_otor, ws, out_frob, st, reg)
                : launch_big<false, false>(p, indptr, indices, values, n_rows, k, this_, other,
                                           otor, ld_otor, ws, out_frob, st, reg);
}

// ---- Gramian  M^T M + reg I  for padded k > 256 --------------------------------------------------
// One wave per (upper tile pair, row slab): f32 MFMA over 4 rows per step, the chain folded into a
// second accumulator every 256 steps (two-level sum, as gramian.hip); the slabs are summed in
// float64 in slab order and rounded once.
namespace big {

constexpr int GRAM_SLABS = 64;

__global__ __launch_bounds__(64) void gramian_big_partial_kernel(const float *__restrict__ m,
                                                                 int64_t n, int KP, int NT,
                                                                 int n_slabs,
                                                                 float *__restrict__ partial)
{
    const int64_t pair = blockIdx.x;
    const int slab = blockIdx.y;
    // pair -> (ti, tj)
    int ti = 0;
    int64_t left = pair;
    while (left >= NT - ti) {
        left -= NT - ti;
        ++ti;
    }
    const int tj = ti + (int)left;
    const int lane = lane_id();
    const int sub = lane & 15, slot = lane >> 4;
    const int64_t per = ((n + n_slabs - 1) / n_slabs + 3) / 4 * 4;
    const int64_t r0 = (int64_t)slab * per, r1 = r0 + per < n ? r0 + per : n;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, tot = f32x4{0.f, 0.f, 0.f, 0.f};
    int steps = 0;
    for (int64_t r = r0; r < r1; r += 4) {
        const int64_t rr = r + slot;
        const bool live = rr < r1;
        const float *mr = m + (live ? rr : r) * KP;
        const float a = live ? mr[16 * ti + sub] : 0.f;
        const float b = live ? mr[16 * tj + sub] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        if (++steps == 256) {
            tot += acc;
            acc = f32x4{0.f, 0.f, 0.f, 0.f};
            steps = 0;
        }
    }
    tot += acc;
    *reinterpret_cast<f32x4 *>(partial + ((size_t)slab * n_pairs(NT) + pair) * 256 + lane * 4) = tot;
}

__global__ void gramian_big_finish_kernel(const float *__restrict__ partial, int NT, int n_slabs,
                                          int k, float reg, float *__restrict__ out, int ld_out)
{
    const int64_t pair = blockIdx.x;
    int ti = 0;
    int64_t left = pair;
    while (left >= NT - ti) {
        left -= NT - ti;
        ++ti;
    }
    const int tj = ti + (int)left;
    const int e = threadIdx.x;  // 256 threads: lane = e >> 2, register = e & 3
    const int lane = e >> 2, r = e & 3;
    const int sub = lane & 15, slot = lane >> 4;
    double s = 0.0;
    for (int sl = 0; sl < n_slabs; ++sl)
        s += (double)partial[((size_t)sl * n_pairs(NT) + pair) * 256 + e];
    const int fi = 16 * ti + 4 * slot + r, fj = 16 * tj + sub;
    if (fi < k && fj < k) {
        float v = (float)s;
        if (fi == fj) v += reg;
        out[(int64_t)fi * ld_out + fj] = v;
        out[(int64_t)fj * ld_out + fi] = v;
    }
}

}  // namespace big

size_t gramian_big_workspace_bytes(int KP)
{
    return (size_t)big::GRAM_SLABS * big::n_pairs(KP / 16) * 256 * sizeof(float);
}

int gramian_big(const float *m, int64_t n, int k, int KP, float reg, float *out, int ld_out,
                float *ws, hipStream_t st)
{
    const int NT = KP / 16;
    int n_slabs = (int)((n + 1023) / 1024);
    if (n_slabs > big::GRAM_SLABS) n_slabs = big::GRAM_SLABS;
    if (n_slabs < 1) n_slabs = 1;
    hipLaunchKernelGGL(big::gramian_big_partial_kernel, dim3((unsigned)big::n_pairs(NT), n_slabs),
                       dim3(64), 0, st, m, n, KP, NT, n_slabs, ws);
    hipLaunchKernelGGL(big::gramian_big_finish_kernel, dim3((unsigned)big::n_pairs(NT)), dim3(256),
                       0, st, ws, NT, n_slabs, k, reg, out, ld_out);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

}  // namespace lk
