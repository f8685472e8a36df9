// spd_inverse.hip -- inverse of the k x k matrix OtOr = O^T O + reg I to float64 accuracy, on
// the device, without a library call or a host round trip.
//
// The Woodbury row kernels (als_wb.hip, als_wb64_kernel) solve short ALS rows as rank-n updates
// of G = OtOr, the SAME matrix for every row of a half-epoch, and need Z = O * G^-1 once per
// half-epoch.  Rounds 1-2 took G^-1 from torch.linalg.cholesky_ex / cholesky_inverse (float64)
// on the caller's side of the C ABI and read the `info` word back: a library call and a host
// synchronisation inside every half-epoch.  Here the whole step stays on the launch stream:
//
//  1. spd_sweep_kernel: one workgroup of 1024 threads holds the padded KP x KP matrix in float32
//     REGISTERS (thread (i, q) owns CW = KP * KP / 1024 consecutive columns of row i) and runs KP
//     steps of the symmetric SWEEP operator -- step j publishes row j through LDS (by symmetry
//     it is also column j: no thread ever needs another thread's registers), d = a_jj,
//         a_il <- a_il - a_ij a_jl / d  (i, l != j),   a_ij, a_ji <- a_ij / d,   a_jj <- -1 / d,
//     after which the registers hold -G^-1 to float32 accuracy, X0.  (A float64 sweep would need
//     the CU's entire register file at KP = 256.)  The pivots are the squared Cholesky pivots:
//     one that is not positive stops the sweep and sets the flag.
//  2. two Newton-Schulz steps in float64, X <- X + X (I - G X): the residual of X0 is
//     ~cond(G) * 2^-24 * c, each step squares it, so two steps reach the float64 floor for
//     cond(G) up to ~1e6; a residual above 1/4 after the first step (no convergence: G is
//     numerically singular) sets the flag instead.  The products are 256-workgroup float64
//     GEMMs of KP^3 multiply-adds -- microseconds.
//  3. the result is rounded ONCE to float32 into a zero-padded [KP x KP] matrix, the item-side
//     operand of the scoring GEMM that forms Z (topk.hip, k-ordered f32 MFMA).
//
// flag != 0 (G not positive definite -- reg = 0 with rank-deficient factors): the Woodbury
// kernels return at once and the dense fallback launch of als_blk.hip takes their rows; decided
// on the device.
#include "common.h"

namespace lk {

// ---- 1. float32 sweep in registers -------------------------------------------------------------
template <int KP>
__global__ __launch_bounds__(1024) void spd_sweep_kernel(const float *__restrict__ a, int lda,
                                                         int k, double *__restrict__ x0,
                                                         int *__restrict__ flag)
{
    constexpr int TPR = 1024 / KP;  // threads per row
    constexpr int CW = KP / TPR;    // columns per thread: 64 at KP = 256, 16 at KP = 128
    __shared__ float prow[KP];
    const int tid = threadIdx.x;
    const int i = tid / TPR, c0 = (tid % TPR) * CW;
    float w[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) {
        const int l = c0 + c;
        // rows / columns beyond k: identity (their pivots are 1, they touch nothing else)
        w[c] = (i < k && l < k) ? a[(int64_t)i * lda + l] : (i == l ? 1.0f : 0.0f);
    }
    int bad = 0;
    for (int j = 0; j < KP; ++j) {
        if (i == j) {
#pragma unroll
            for (int c = 0; c < CW; ++c) prow[c0 + c] = w[c];
        }
        __syncthreads();
        const float d = prow[j];
        if (!(d > 0.0f)) {  // the same value in every thread: a uniform exit
            bad = j + 1;
            break;
        }
        const float rd = 1.0f / d;
        const float t = prow[i] * rd;  // a_ij / d  (a_ij = a_ji: row j is column j)
        if (i == j) {
#pragma unroll
            for (int c = 0; c < CW; ++c) w[c] = (c0 + c == j) ? -rd : prow[c0 + c] * rd;
        } else {
#pragma unroll
            for (int c = 0; c < CW; ++c)
                w[c] = (c0 + c == j) ? t : __builtin_fmaf(-t, prow[c0 + c], w[c]);
        }
        __syncthreads();  // everybody has read row j before step j + 1 overwrites it
    }
    if (tid == 0) *flag = bad;
#pragma unroll
    for (int c = 0; c < CW; ++c) x0[(int64_t)i * KP + c0 + c] = bad ? 0.0 : (double)(-w[c]);
}

// ---- 2. Newton-Schulz in float64 ---------------------------------------------------------------
// One 16 x 16 tile of C = alpha * A B + beta_I * I (+ D) per workgroup, float64 accumulation.
// MODE 0: R = I - G X   (A = G, float32 [k x k] padded with the identity; B = X)
// MODE 1: Xn = X + X R  (A = X, B = R, D = X); LAST: also the float32, zero-padded result and the
//         convergence check on R.
template <int KP, int MODE, bool LAST>
__global__ __launch_bounds__(256) void spd_ns_kernel(const float *__restrict__ g, int ldg, int k,
                                                     const double *__restrict__ x,
                                                     const double *__restrict__ r,
                                                     double *__restrict__ out,
                                                     float *__restrict__ out32,
                                                     int *__restrict__ flag)
{
    __shared__ double sa[16][KP + 1];  // rows of A of this tile
    __shared__ double sb[KP][16 + 1];  // columns of B of this tile
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    const int i0 = (blockIdx.x / (KP / 16)) * 16, j0 = (blockIdx.x % (KP / 16)) * 16;
    for (int e = tid; e < 16 * KP; e += 256) {
        const int rr = e / KP, cc = e % KP;  // A[i0 + rr][cc]
        const int i = i0 + rr;
        double v;
        if (MODE == 0)
            v = (i < k && cc < k) ? (double)g[(int64_t)i * ldg + cc] : (i == cc ? 1.0 : 0.0);
        else
            v = x[(int64_t)i * KP + cc];
        sa[rr][cc] = v;
        const int kk = e / 16, jj = e % 16;  // B[kk][j0 + jj]
        sb[kk][jj] = (MODE == 0 ? x : r)[(int64_t)kk * KP + j0 + jj];
    }
    __syncthreads();
    double acc = 0.0;
#pragma unroll 8
    for (int kk = 0; kk < KP; ++kk) acc = __builtin_fma(sa[ti][kk], sb[kk][tj], acc);
    const int i = i0 + ti, j = j0 + tj;
    if (MODE == 0) {
        const double res = (i == j ? 1.0 : 0.0) - acc;
        out[(int64_t)i * KP + j] = res;
        // no convergence (|I - G X0| is not small): G is numerically singular for this path
        if (!(__builtin_fabs(res) < 0.25)) atomicMax(flag, KP + 1);
    } else {
        const double xn = x[(int64_t)i * KP + j] + acc;
        out[(int64_t)i * KP + j] = xn;
        if (LAST) out32[(int64_t)i * KP + j] = (*flag == 0 && i < k && j < k) ? (float)xn : 0.f;
    }
}

template <int KP>
static int spd_inverse_launch(const float *a, int lda, int k, float *out, int *flag, double *ws,
                              hipStream_t st)
{
    double *x = ws, *r = ws + (size_t)KP * KP, *xn = ws + (size_t)2 * KP * KP;
    const dim3 tiles((KP / 16) * (KP / 16)), blk(256);
    hipLaunchKernelGGL(spd_sweep_kernel<KP>, dim3(1), dim3(1024), 0, st, a, lda, k, x, flag);
    hipLaunchKernelGGL((spd_ns_kernel<KP, 0, false>), tiles, blk, 0, st, a, lda, k, x, nullptr, r,
                       nullptr, flag);
    hipLaunchKernelGGL((spd_ns_kernel<KP, 1, false>), tiles, blk, 0, st, a, lda, k, x, r, xn,
                       nullptr, flag);
    hipLaunchKernelGGL((spd_ns_kernel<KP, 0, false>), tiles, blk, 0, st, a, lda, k, xn, nullptr, r,
                       nullptr, flag);
    hipLaunchKernelGGL((spd_ns_kernel<KP, 1, true>), tiles, blk, 0, st, a, lda, k, xn, r, x, out,
                       flag);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

size_t spd_inverse_workspace_bytes(int KP) { return (size_t)3 * KP * KP * sizeof(double); }

// out[KP x KP] (zero padded, float32) = (a[k x k])^-1 to float64 accuracy; *flag = 0, or != 0 when
// a is not (numerically) positive definite -- then `out` is all zeros.  KP = 128 or 256; `ws`:
// spd_inverse_workspace_bytes(KP) bytes, 8-byte aligned.
int spd_inverse(const float *a, int lda, int k, int KP, float *out, int *flag, void *ws,
                hipStream_t st)
{
    if (KP == 256) return spd_inverse_launch<256>(a, lda, k, out, flag, static_cast<double *>(ws), st);
    if (KP == 128) return spd_inverse_launch<128>(a, lda, k, out, flag, static_cast<double *>(ws), st);
    set_error("spd_inverse: unsupported padded size %d", KP);
    return LK_E_INVALID;
}

}  // namespace lk

extern "C" size_t lk_spd_inverse_workspace_bytes(int32_t k)
{
    const int KP = lk_padded_dim(k);
    return (KP == 128 || KP == 256) ? lk::spd_inverse_workspace_bytes(KP) : 0;
}

// Test / diagnostic entry (the half-epoch calls lk::spd_inverse itself): d_out [KP x KP] floats,
// d_flag one int, d_ws lk_spd_inverse_workspace_bytes(k) bytes -- device pointers.
extern "C" int lk_spd_inverse(const float *d_a, int32_t lda, int32_t k, float *d_out,
                              int32_t *d_flag, void *d_ws, void *stream)
{
    const int KP = lk_padded_dim(k);
    LK_REQUIRE(KP == 128 || KP == 256, "lk_spd_inverse: padded size %d (k = %d) not served", KP, k);
    LK_REQUIRE(d_a && d_out && d_flag && d_ws && lda >= k, "lk_spd_inverse: bad arguments");
    return lk::spd_inverse(d_a, lda, k, KP, d_out, d_flag, d_ws, lk::as_stream(stream));
}
