// als_cg.hip -- implicit-ALS half-epoch, Jacobi-preconditioned conjugate gradient, gfx950.
//
// Same contract as the Cholesky kernel (src/accel/als/implicit.rs:35-125): per CSR row
//     A = OtOr + sum_j v_j q_j q_j^T,   y = sum_j (v_j + 1) q_j,   A x = y,
// but A is never formed: A p = OtOr p + sum_j v_j q_j (q_j . p)  (the north star's "per-user
// CG step").  This is the solver for k > 64 (normal matrices of 64 KiB / 256 KiB do not fit
// the per-wave register/LDS budget of the exact path); it is tolerance-terminated
// (||r|| <= tol ||y||, at most max_iter iterations), warm-started from the previous
// factor row, so its result converges to the exact solve the reference computes.
//
// One workgroup (4 waves) per row.  Every wave keeps an identical copy of the CG vectors
// (x, r, z, p, diagonal), lane l holding the FPL = KP/64 contiguous features l*FPL..; the
// items of the row are dealt to the waves (item j -> wave j mod 4): one coalesced 4*KP-byte
// gather per item and iteration, a wave-shuffle reduction for q_j . p, an axpy into the
// wave's partial A p.  The dense term is split the same way (wave w takes rows
// w*KP/4 .. of OtOr, read coalesced from L2).  The four partial vectors meet in LDS once
// per iteration and are summed in wave order, so all waves stay bit-identical and the
// result is deterministic.
//
// Roofline: HBM/L2 bound -- algorithmic bytes per iteration nnz*(4k + 8) + rows*4k^2/…;
// flops T*(nnz*4k + rows*2k^2).  First version: no LDS residency of the
// gathered rows across iterations and one row per workgroup (round-2 work: batch rows so
// that OtOr . P becomes an MFMA GEMM, keep short rows' gathers in LDS).
#include "als_plan.h"
#include "common.h"

namespace lk {

template <int FPL>
struct Vec {
    float v[FPL];
};

template <int FPL>
__device__ __forceinline__ Vec<FPL> vload(const float *p)
{
    Vec<FPL> r;
    if constexpr (FPL == 1) {
        r.v[0] = p[0];
    } else if constexpr (FPL == 2) {
        f32x2 t = *reinterpret_cast<const f32x2 *>(p);
        r.v[0] = t.x;
        r.v[1] = t.y;
    } else {
        f32x4 t = *reinterpret_cast<const f32x4 *>(p);
        r.v[0] = t.x;
        r.v[1] = t.y;
        r.v[2] = t.z;
        r.v[3] = t.w;
    }
    return r;
}

template <int FPL>
__device__ __forceinline__ float vdot(const Vec<FPL> &a, const Vec<FPL> &b)
{
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < FPL; ++c) s = fmaf(a.v[c], b.v[c], s);
    return wave_sum(s);
}

template <int KP, bool IS64>
__global__ __launch_bounds__(256) void als_cg_kernel(
    const typename IndPtr<IS64>::type *__restrict__ indptr, const int32_t *__restrict__ indices,
    const float *__restrict__ values, const int32_t *__restrict__ order, int64_t n_rows,
    const float *__restrict__ other, float *__restrict__ this_, const float *__restrict__ otor,
    int ld_otor, int k, float tol, int max_iter, float *__restrict__ row_delta,
    int *__restrict__ status, TaskCtlDev ctl)
{
    constexpr int FPL = KP / 64;
    constexpr bool OTOR_LDS = KP <= 128;  // 16 / 64 KiB: loaded once per (persistent) workgroup
    constexpr int GB = 8;                 // items gathered per wave and batch
    __shared__ float part[4][KP];
    extern __shared__ __attribute__((aligned(16))) float otor_s[];  // OTOR_LDS: KP*KP floats
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const int f0 = lane * FPL;  // first feature of this lane
    if (OTOR_LDS) {  // zero padded to KP x KP
        for (int e = threadIdx.x; e < KP * KP; e += 256) {
            const int g = e / KP, f = e % KP;
            otor_s[e] = (g < k && f < k) ? otor[(int64_t)g * ld_otor + f] : 0.f;
        }
        __syncthreads();
    }

    __shared__ int s_cancel;
    for (int64_t t = blockIdx.x; t < n_rows; t += gridDim.x) {
        if (ctl.d_cancel) {  // AccelTask.cancel: rows not started yet are skipped
            if (threadIdx.x == 0) s_cancel = ctl_cancelled(ctl, (blockIdx.x & 31) == 0) ? 1 : 0;
            __syncthreads();
            const int c = s_cancel;
            __syncthreads();
            if (c) return;
        }
        const int row = order[t];
        const int64_t beg = indptr[row], end = indptr[row + 1];
        float *xrow = this_ + (int64_t)row * KP;
        if (end == beg) {  // implicit.rs:98-101
            if (wave == 0)
#pragma unroll
                for (int c = 0; c < FPL; ++c) xrow[f0 + c] = 0.f;
            if (threadIdx.x == 0) row_delta[row] = 0.f;
            if (ctl.d_done && threadIdx.x == 0) ctl_advance(ctl, 1);
            continue;
        }
        // A p accumulated over this wave's share of the items and of the OtOr rows,
        // then combined across the four waves (fixed order)
        auto apply = [&](const Vec<FPL> &p, Vec<FPL> &out) {
            Vec<FPL> u;
#pragma unroll
            for (int c = 0; c < FPL; ++c) u.v[c] = 0.f;
            // items: wave w takes batches w, w+4, ... of GB consecutive entries; the GB
            // gathers of a batch are in flight together and their GB dot-product
            // reductions are interleaved (independent shuffle chains)
            for (int64_t e0 = beg + (int64_t)wave * GB; e0 < end; e0 += 4 * GB) {
                Vec<FPL> q[GB];
                float dotp[GB], val[GB];
#pragma unroll
                for (int j = 0; j < GB; ++j) {
                    const bool in = e0 + j < end;
                    const int64_t it = in ? indices[e0 + j] : indices[beg];
                    val[j] = in ? values[e0 + j] : 0.f;
                    q[j] = vload<FPL>(other + it * KP + f0);
                }
#pragma unroll
                for (int j = 0; j < GB; ++j) {
                    float sacc = 0.f;
#pragma unroll
                    for (int c = 0; c < FPL; ++c) sacc = fmaf(q[j].v[c], p.v[c], sacc);
                    dotp[j] = sacc;
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1)
#pragma unroll
                    for (int j = 0; j < GB; ++j) dotp[j] += __shfl_xor(dotp[j], off, 64);
#pragma unroll
                for (int j = 0; j < GB; ++j) {
                    const float coef = val[j] * dotp[j];
#pragma unroll
                    for (int c = 0; c < FPL; ++c) u.v[c] = fmaf(coef, q[j].v[c], u.v[c]);
                }
            }
            for (int g = wave * (KP / 4); g < (wave + 1) * (KP / 4); ++g) {
                if (g >= k) break;
                // p[g] lives in lane g / FPL, component g % FPL
                float pg = 0.f;
#pragma unroll
                for (int c = 0; c < FPL; ++c)
                    if ((g % FPL) == c) pg = bcast(p.v[c], g / FPL);
                // row g of the symmetric OtOr == column g; pad features contribute nothing
                if (OTOR_LDS) {
                    const Vec<FPL> o = vload<FPL>(&otor_s[g * KP + f0]);
#pragma unroll
                    for (int c = 0; c < FPL; ++c) u.v[c] = fmaf(pg, o.v[c], u.v[c]);
                } else {
#pragma unroll
                    for (int c = 0; c < FPL; ++c) {
                        const int f = f0 + c;
                        const float o = (f < k) ? otor[(int64_t)g * ld_otor + f] : 0.f;
                        u.v[c] = fmaf(pg, o, u.v[c]);
                    }
                }
            }
            __syncthreads();  // previous readers of `part` are done
#pragma unroll
            for (int c = 0; c < FPL; ++c) part[wave][f0 + c] = u.v[c];
            __syncthreads();
#pragma unroll
            for (int c = 0; c < FPL; ++c)
                out.v[c] = ((part[0][f0 + c] + part[1][f0 + c]) + part[2][f0 + c]) +
                           part[3][f0 + c];
        };

        // y = sum (v+1) q  and the Jacobi diagonal  d = diag(OtOr) + sum v q^2
        Vec<FPL> y, d;
        {
            Vec<FPL> yw, dw;
#pragma unroll
            for (int c = 0; c < FPL; ++c) yw.v[c] = dw.v[c] = 0.f;
            for (int64_t e0 = beg + (int64_t)wave * GB; e0 < end; e0 += 4 * GB) {
                Vec<FPL> q[GB];
                float val[GB];
#pragma unroll
                for (int j = 0; j < GB; ++j) {
                    const bool in = e0 + j < end;
                    const int64_t it = in ? indices[e0 + j] : indices[beg];
                    val[j] = in ? values[e0 + j] : -1.0f;  // (v + 1) = 0 and v q q = -q q ...
                    q[j] = vload<FPL>(other + it * KP + f0);
                    if (!in)
#pragma unroll
                        for (int c = 0; c < FPL; ++c) q[j].v[c] = 0.f;  // ... of a zero row
                }
#pragma unroll
                for (int j = 0; j < GB; ++j)
#pragma unroll
                    for (int c = 0; c < FPL; ++c) {
                        yw.v[c] = fmaf(val[j] + 1.0f, q[j].v[c], yw.v[c]);
                        dw.v[c] = fmaf(val[j] * q[j].v[c], q[j].v[c], dw.v[c]);
                    }
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < FPL; ++c) part[wave][f0 + c] = yw.v[c];
            __syncthreads();
#pragma unroll
            for (int c = 0; c < FPL; ++c)
                y.v[c] = ((part[0][f0 + c] + part[1][f0 + c]) + part[2][f0 + c]) +
                         part[3][f0 + c];
            __syncthreads();
#pragma unroll
            for (int c = 0; c < FPL; ++c) part[wave][f0 + c] = dw.v[c];
            __syncthreads();
#pragma unroll
            for (int c = 0; c < FPL; ++c) {
                const int f = f0 + c;
                const float od = (f < k) ? otor[(int64_t)f * ld_otor + f] : 1.0f;
                d.v[c] = od + (((part[0][f] + part[1][f]) + part[2][f]) + part[3][f]);
            }
        }
        Vec<FPL> x, xold, r, z, p, ap;
#pragma unroll
        for (int c = 0; c < FPL; ++c) xold.v[c] = x.v[c] = xrow[f0 + c];  // warm start
        apply(x, ap);
        const float ynorm2 = vdot<FPL>(y, y);
#pragma unroll
        for (int c = 0; c < FPL; ++c) {
            r.v[c] = y.v[c] - ap.v[c];
            z.v[c] = r.v[c] / d.v[c];
            p.v[c] = z.v[c];
        }
        float rz = vdot<FPL>(r, z);
        float rr = vdot<FPL>(r, r);
        const float stop = tol * tol * ynorm2;
        int it = 0;
        bool bad = !(ynorm2 == ynorm2);
        while (it < max_iter && rr > stop && !bad) {
            apply(p, ap);
            const float pap = vdot<FPL>(p, ap);
            if (!(pap > 0.f)) {  // not positive definite along p
                bad = true;
                break;
            }
            const float alpha = rz / pap;
#pragma unroll
            for (int c = 0; c < FPL; ++c) {
                x.v[c] = fmaf(alpha, p.v[c], x.v[c]);
                r.v[c] = fmaf(-alpha, ap.v[c], r.v[c]);
                z.v[c] = r.v[c] / d.v[c];
            }
            const float rz_new = vdot<FPL>(r, z);
            rr = vdot<FPL>(r, r);
            const float beta = rz_new / rz;
            rz = rz_new;
#pragma unroll
            for (int c = 0; c < FPL; ++c) p.v[c] = fmaf(beta, p.v[c], z.v[c]);
            ++it;
        }
        float dd = 0.f;
#pragma unroll
        for (int c = 0; c < FPL; ++c) {
            const float df = (f0 + c < k) ? x.v[c] - xold.v[c] : 0.f;
            dd = fmaf(df, df, dd);
            bad = bad || !(fabsf(x.v[c]) <= 3.0e38f);
        }
        dd = wave_sum(dd);
        if (wave == 0) {
#pragma unroll
            for (int c = 0; c < FPL; ++c) xrow[f0 + c] = (f0 + c < k) ? x.v[c] : 0.f;
            if (lane == 0) row_delta[row] = dd;
        }
        if (__any(bad) && threadIdx.x == 0) atomicCAS(status, 0, row + 1);
        if (ctl.d_done && threadIdx.x == 0) ctl_advance(ctl, 1);
        __syncthreads();
    }
}

template <int KP, bool IS64>
static int launch_cg(const lk_als_plan *p, const void *indptr, const int32_t *indices,
                     const float *values, int64_t n_rows, int k, float *this_,
                     const float *other, const float *otor, int ld_otor, char *ws,
                     float *out_frob, hipStream_t st)
{
    using IT = typename IndPtr<IS64>::type;
    int *status = reinterpret_cast<int *>(ws + p->off_status);
    float *row_delta = reinterpret_cast<float *>(ws + p->off_delta);
    float *partial = reinterpret_cast<float *>(ws + p->off_partial);
    LK_HIP_CHECK(hipMemsetAsync(status, 0, 64, st));
    if (p->ctl) {
        LK_HIP_CHECK(hipMemsetAsync(row_delta, 0, (size_t)n_rows * sizeof(float), st));
        int rc = ctl_begin(p->ctl, n_rows, n_rows, st);
        if (rc != LK_OK) return rc;
    }
    const int max_iter = p->cg_max_iter > 0 ? p->cg_max_iter : k;
    if (n_rows > 0) {
        int64_t blocks = n_rows < 256 * 8 ? n_rows : 256 * 8;
        const size_t lds = KP <= 128 ? (size_t)KP * KP * sizeof(float) : 0;  // OtOr resident
        auto kern = als_cg_kernel<KP, IS64>;
        if (lds > 0)
            LK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st,
                           static_cast<const IT *>(indptr), indices, values, p->d_order, n_rows,
                           other, this_, otor, ld_otor, k, p->cg_tol, max_iter, row_delta,
                           status, p->ctl ? p->ctl->dev() : TaskCtlDev{});
    }
    return launch_delta_reduce(row_delta, n_rows, partial, out_frob, st);
}

int als_cg_half_epoch(const lk_als_plan *p, const void *indptr, int is64, const int32_t *indices,
                      const float *values, int64_t n_rows, int k, float *this_, int ld_this,
                      const float *other, int ld_other, const float *otor, int ld_otor, char *ws,
                      float *out_frob, hipStream_t st)
{
    (void)ld_this;
    (void)ld_other;
#define LK_CG_CASE(KPV)                                                                     \
    return is64 ? launch_cg<KPV, true>(p, indptr, indices, values, n_rows, k, this_, other, \
                                       otor, ld_otor, ws, out_frob, st)                     \
                : launch_cg<KPV, false>(p, indptr, indices, values, n_rows, k, this_, other, \
                                        otor, ld_otor, ws, out_frob, st)
    switch (p->KP) {
        case 64: LK_CG_CASE(64);
        case 128: LK_CG_CASE(128);
        case 256: LK_CG_CASE(256);
    }
#undef LK_CG_CASE
    set_error("CG solver: unsupported padded embedding size %d (needs k > 32)", p->KP);
    return LK_E_INVALID;
}

}  // namespace lk
