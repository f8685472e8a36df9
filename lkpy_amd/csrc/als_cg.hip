// als_cg.hip -- implicit-ALS half-epoch, Jacobi-preconditioned conjugate gradient, gfx950.
//
// Same contract as the Cholesky kernels (src/accel/als/implicit.rs:35-125): per CSR row
//     A = OtOr + sum_j v_j q_j q_j^T,   y = sum_j (v_j + 1) q_j,   A x = y,
// but A is never formed: A p = OtOr p + sum_j v_j q_j (q_j . p)  (the north star's "per-user
// CG step").  Tolerance-terminated (||r|| <= tol ||y||, at most max_iter iterations), warm-started
// from the previous factor row, so the result converges to the exact solve the reference computes.
// An OPTION (solver = "cg"): the exact kernels are the default at every k.
//
// Lane l of a wave holds the FPL = KP/64 contiguous features l*FPL.. of every CG vector (x, r, z,
// p, 1/diagonal) and of every gathered factor row.  A row's gathered factor rows stay in
// REGISTERS over the iterations (RI = 64/FPL rows = 64 VGPRs per wave), so its entries are read
// once per half-epoch.  Two instances:
//   NW = 4  one workgroup per row: the entries and the rows of OtOr are split over the four waves,
//           the four partial A p meet in LDS once per iteration and are summed in wave order (all
//           waves keep bit-identical vectors; deterministic);  rows of up to 4 RI entries, longer
//           ones gather their tail again in every iteration;
//   NW = 1  one WAVE per row, four rows per workgroup: no barrier, no LDS exchange; rows of up to
//           RI entries.
// Rows longer than 4 RI entries (16384 / KP) are handed to the exact kernels by
// als_cg_half_epoch (see there).  The dot products q_j . p of 8 entries are reduced together
// (`cg_reduce8`), the CG scalars by DPP row sums + readlane (`cg_wave_sum`); OtOr is LDS resident
// for KP <= 128.
//
// Roofline: HBM/L2 bound -- flops T*(nnz*4k + rows*2k^2), bytes nnz*(4k + 8) + rows*8k
// (SURVEY.md section 8d).
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

template <int CTRL>
__device__ __forceinline__ float cg_dpp(float x)
{
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
// the CG scalars (p.Ap, r.z, r.r): four DPP steps sum a row of 16 lanes, the four row sums are
// read as scalars -- ~60 cycles of dependent latency instead of six LDS-pipe exchanges (~600)
// on the serial path of every iteration.  Same value in every lane and in every wave.
__device__ __forceinline__ float cg_wave_sum(float x)
{
    x += cg_dpp<0xB1>(x);   // quad_perm [1,0,3,2]
    x += cg_dpp<0x4E>(x);   // quad_perm [2,3,0,1]
    x += cg_dpp<0x141>(x);  // row_half_mirror
    x += cg_dpp<0x140>(x);  // row_mirror
    return (bcast(x, 0) + bcast(x, 16)) + (bcast(x, 32) + bcast(x, 48));
}
template <int FPL>
__device__ __forceinline__ float vdot(const Vec<FPL> &a, const Vec<FPL> &b)
{
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < FPL; ++c) s = fmaf(a.v[c], b.v[c], s);
    return cg_wave_sum(s);
}

__device__ __forceinline__ int64_t cg_uniform64(int64_t v)
{
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
    return (int64_t)(((unsigned long long)hi << 32) | lo);
}

// ---- 8 dot products reduced over the wave at once ------------------------------------------
// a[j] = this lane's share of q_j . p for 8 items.  Instead of 8 separate 6-step butterflies
// (48 exchanges) the values are folded while they are summed: step A pairs lanes L, L^1 and
// leaves each with 4 of the 8 items, step B (L^2) with 2, step C (L^4) with one; then the 8
// lanes that hold the same item are summed across the wave (8, 16, 32).  10 exchanges, 7 of them
// DPP (quad_perm / row_ror: VALU, no LDS pipe).  Returns the complete dot product of item
// rev3(lane & 7) (bit-reversed: lane bit 0 chose item bit 2, ...).
__device__ __forceinline__ float cg_reduce8(const float (&a)[8], int lane)
{
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    float b[4], c[2];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float keep = b0 ? a[m + 4] : a[m], send = b0 ? a[m] : a[m + 4];
        b[m] = keep + cg_dpp<0xB1>(send);  // quad_perm [1,0,3,2]: lane ^ 1
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const float keep = b1 ? b[m + 2] : b[m], send = b1 ? b[m] : b[m + 2];
        c[m] = keep + cg_dpp<0x4E>(send);  // quad_perm [2,3,0,1]: lane ^ 2
    }
    const float keep = b2 ? c[1] : c[0], send = b2 ? c[0] : c[1];
    float d = keep + __shfl_xor(send, 4, 64);
    d += cg_dpp<0x128>(d);  // row_ror:8 = lane ^ 8 inside the row of 16
    d += __shfl_xor(d, 16, 64);
    d += __shfl_xor(d, 32, 64);
    return d;
}
// item j of a group of 8 is finished in the lanes with (lane & 7) == CG_REV3(j)
#define CG_REV3(j) ((((j) & 1) << 2) | ((j) & 2) | (((j) & 4) >> 2))

// NW = waves that share one row: 4 (a workgroup per row; rows of up to 4 * RI resident entries,
// partial vectors combined in LDS with two barriers per iteration) or 1 (a WAVE per row, four rows
// per workgroup, for rows of at most RI entries: no barrier, no LDS exchange).
template <int KP, bool IS64, int NW>
__global__ __launch_bounds__(256, (NW == 1 && KP <= 128) ? 3 : 2) void als_cg_kernel(
    const typename IndPtr<IS64>::type *__restrict__ indptr, const int32_t *__restrict__ indices,
    const float *__restrict__ values, const int32_t *__restrict__ order, int64_t n_rows,
    const float *__restrict__ other, float *__restrict__ this_, const float *__restrict__ otor,
    int ld_otor, int k, float tol, int max_iter, float *__restrict__ row_delta,
    int *__restrict__ status, TaskCtlDev ctl, int64_t t_begin, int64_t t_end)
{
    static_assert(NW == 4 || NW == 1, "NW");
    constexpr int FPL = KP / 64;
    constexpr bool OTOR_LDS = KP <= 128;  // 16 / 64 KiB: loaded once per (persistent) workgroup
    constexpr int GB = 8;                 // items per reduction group
    constexpr int RI = 64 / FPL;          // items a wave keeps in registers (64 VGPRs)
    constexpr int NG = RI / GB;
    constexpr int TB = FPL <= 2 ? 16 : 8;  // tail entries a wave gathers per batch
    __shared__ float part[4][KP];
    extern __shared__ __attribute__((aligned(16))) float otor_s[];  // OTOR_LDS: KP*KP floats
    const int wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = NW == 4 ? wave_in_wg : 0;  // this wave's place in its row's team
    const bool lead = NW == 1 || wave_in_wg == 0;
    const int lane = lane_id();
    const int f0 = lane * FPL;  // first feature of this lane
    auto team_sync = [&]() {
        if constexpr (NW == 4) __syncthreads();
    };
    const int myj = CG_REV3(lane & 7);  // the item of a group whose total this lane ends up with
    if (OTOR_LDS) {  // zero padded to KP x KP
        for (int e = threadIdx.x; e < KP * KP; e += 256) {
            const int g = e / KP, f = e % KP;
            otor_s[e] = (g < k && f < k) ? otor[(int64_t)g * ld_otor + f] : 0.f;
        }
        __syncthreads();
    }

    __shared__ int s_cancel;
    // tasks [t_begin, n_rows) of the longest-first order (the chunked rows before t_begin were
    // solved by the exact kernels: als_cg_half_epoch)
    (void)n_rows;
    for (int64_t t = t_begin + (NW == 4 ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * 4 + wave_in_wg);
         t < t_end; t += (NW == 4 ? (int64_t)gridDim.x : (int64_t)gridDim.x * 4)) {
        if (NW == 4 && ctl.d_cancel) {  // AccelTask.cancel: rows not started yet are skipped
            if (threadIdx.x == 0) s_cancel = ctl_cancelled(ctl, (blockIdx.x & 31) == 0) ? 1 : 0;
            __syncthreads();
            const int c = s_cancel;
            __syncthreads();
            if (c) return;
        }
        // wave-uniform on purpose: the entry numbers and item numbers below become scalar loads
        // and the gathers `scalar base + lane offset` (no 64-bit address registers per item)
        const int row = __builtin_amdgcn_readfirstlane(order[t]);
        const int64_t beg = cg_uniform64(indptr[row]), end = cg_uniform64(indptr[row + 1]);
        float *xrow = this_ + (int64_t)row * KP;
        if (end == beg) {  // implicit.rs:98-101
            if (lead)
#pragma unroll
                for (int c = 0; c < FPL; ++c) xrow[f0 + c] = 0.f;
            if (lead && lane == 0) row_delta[row] = 0.f;
            if (ctl.d_done && lead && lane == 0) ctl_advance(ctl, 1);
            continue;
        }
        // The row's gathered factor rows stay in REGISTERS for the whole solve: the first
        // 4 * RI entries are dealt to the waves in equal contiguous shares (a multiple of 8),
        // wave w keeps its share as qres[i] (lane = feature(s)); only the entries past 4 * RI
        // of a long row are gathered again in every iteration (`stream`).
        const int64_t n = end - beg;
        const int share = n >= NW * RI ? RI : (int)((n + NW * GB - 1) / (NW * GB)) * GB;
        const int64_t rbeg = beg + (int64_t)wave * share;
        int nres = (int)(end - rbeg < share ? end - rbeg : share);
        nres = __builtin_amdgcn_readfirstlane(nres < 0 ? 0 : nres);
        const int64_t sbeg = beg + NW * (int64_t)share;  // streamed tail (empty unless n > NW RI)
        Vec<FPL> qres[RI];
        float vg[NG];  // lane: the value of item g * 8 + myj of the share (0 past its end)
        // one coalesced load of the share's item numbers (lane i: item i), handed out by readlane
        const int myitem = indices[lane < nres ? rbeg + lane : beg];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            vg[g] = 0.f;
            if (g * GB < nres) {
                vg[g] = (g * GB + myj < nres) ? values[rbeg + g * GB + myj] : 0.f;
#pragma unroll
                for (int j = 0; j < GB; ++j) {
                    const int i = g * GB + j;
                    const int64_t it = __builtin_amdgcn_readlane(myitem, i);
                    qres[i] = vload<FPL>(other + it * KP + f0);
                    if (i >= nres)  // (item `beg` was read: finite, then dropped)
#pragma unroll
                        for (int c = 0; c < FPL; ++c) qres[i].v[c] = 0.f;
                }
            } else {  // (apply works on pairs of groups: an unused partner is zero rows)
#pragma unroll
                for (int j = 0; j < GB; ++j)
#pragma unroll
                    for (int c = 0; c < FPL; ++c) qres[g * GB + j].v[c] = 0.f;
            }
        }
        // A p: this wave's items and its quarter of the OtOr rows, then the four partial
        // vectors are combined in LDS (fixed order: all waves keep bit-identical vectors)
        auto apply = [&](const Vec<FPL> &p, Vec<FPL> &out) {
            Vec<FPL> u;
#pragma unroll
            for (int c = 0; c < FPL; ++c) u.v[c] = 0.f;
            // (two groups per branch: their reductions -- dependent DPP / LDS-pipe chains --
            // interleave)
#pragma unroll
            for (int g2 = 0; g2 < NG; g2 += 2) {
                if (g2 * GB < nres) {
                    float a[2][GB];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int j = 0; j < GB; ++j) {
                            float sacc = 0.f;
#pragma unroll
                            for (int c = 0; c < FPL; ++c)
                                sacc = fmaf(qres[(g2 + h) * GB + j].v[c], p.v[c], sacc);
                            a[h][j] = sacc;
                        }
                    float coefv[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) coefv[h] = vg[g2 + h] * cg_reduce8(a[h], lane);
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int j = 0; j < GB; ++j) {
                            const float coef = bcast(coefv[h], CG_REV3(j));
#pragma unroll
                            for (int c = 0; c < FPL; ++c)
                                u.v[c] = fmaf(coef, qres[(g2 + h) * GB + j].v[c], u.v[c]);
                        }
                }
            }
            // the tail of a long row: wave w takes batches w, w + 4, ... of TB = 16 entries (two
            // reduction groups; the 16 gathers of a batch are in flight together)
            for (int64_t e0 = sbeg + (int64_t)wave * TB; e0 < end; e0 += NW * TB) {
                Vec<FPL> q[TB];
                const int tailitem = indices[(lane < TB && e0 + lane < end) ? e0 + lane : beg];
                float vmine[TB / GB];
#pragma unroll
                for (int h = 0; h < TB / GB; ++h)
                    vmine[h] = (e0 + h * GB + myj < end) ? values[e0 + h * GB + myj] : 0.f;
#pragma unroll
                for (int j = 0; j < TB; ++j) {
                    const int64_t it = __builtin_amdgcn_readlane(tailitem, j);
                    q[j] = vload<FPL>(other + it * KP + f0);
                }
#pragma unroll
                for (int h = 0; h < TB / GB; ++h) {
                    float a[GB];
#pragma unroll
                    for (int j = 0; j < GB; ++j) {
                        float sacc = 0.f;
#pragma unroll
                        for (int c = 0; c < FPL; ++c) sacc = fmaf(q[h * GB + j].v[c], p.v[c], sacc);
                        a[j] = sacc;
                    }
                    const float coefv = vmine[h] * cg_reduce8(a, lane);  // 0 past the end of the row
#pragma unroll
                    for (int j = 0; j < GB; ++j) {
                        const float coef = bcast(coefv, CG_REV3(j));
#pragma unroll
                        for (int c = 0; c < FPL; ++c)
                            u.v[c] = fmaf(coef, q[h * GB + j].v[c], u.v[c]);
                    }
                }
            }
            // (LDS copy: zero padded, so no early exit and the loop unrolls into batched reads)
#pragma unroll 8
            for (int gi = 0; gi < KP / NW; ++gi) {
                const int g = wave * (KP / NW) + gi;
                if (!OTOR_LDS && g >= k) continue;
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
            if constexpr (NW == 1) {
                out = u;
            } else {
                __syncthreads();  // previous readers of `part` are done
#pragma unroll
                for (int c = 0; c < FPL; ++c) part[wave][f0 + c] = u.v[c];
                __syncthreads();
#pragma unroll
                for (int c = 0; c < FPL; ++c)
                    out.v[c] = ((part[0][f0 + c] + part[1][f0 + c]) + part[2][f0 + c]) +
                               part[3][f0 + c];
            }
        };

        // y = sum (v+1) q  and the Jacobi diagonal  d = diag(OtOr) + sum v q^2
        Vec<FPL> y, d;
        {
            Vec<FPL> yw, dw;
#pragma unroll
            for (int c = 0; c < FPL; ++c) yw.v[c] = dw.v[c] = 0.f;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g * GB < nres) {
#pragma unroll
                    for (int j = 0; j < GB; ++j) {
                        // padding entries: a zero row, so the (v + 1) = 1 adds nothing
                        const float val = bcast(vg[g], CG_REV3(j));
#pragma unroll
                        for (int c = 0; c < FPL; ++c) {
                            const float qc = qres[g * GB + j].v[c];
                            yw.v[c] = fmaf(val + 1.0f, qc, yw.v[c]);
                            dw.v[c] = fmaf(val * qc, qc, dw.v[c]);
                        }
                    }
                }
            }
            for (int64_t e0 = sbeg + (int64_t)wave * GB; e0 < end; e0 += NW * GB) {
                Vec<FPL> q[GB];
                float val[GB];
                const bool mine = lane < GB && e0 + lane < end;
                const int tailitem = indices[mine ? e0 + lane : beg];
                const float tailval = mine ? values[e0 + lane] : -1.0f;
#pragma unroll
                for (int j = 0; j < GB; ++j) {
                    const bool in = e0 + j < end;
                    const int64_t it = __builtin_amdgcn_readlane(tailitem, j);
                    val[j] = bcast(tailval, j);  // past the end: (v + 1) = 0 and v q q = -q q ...
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
            if constexpr (NW == 1) {
#pragma unroll
                for (int c = 0; c < FPL; ++c) {
                    const int f = f0 + c;
                    const float od = (f < k) ? otor[(int64_t)f * ld_otor + f] : 1.0f;
                    y.v[c] = yw.v[c];
                    d.v[c] = od + dw.v[c];
                }
            } else {
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
        }
        Vec<FPL> dinv;  // Jacobi preconditioner, inverted once
#pragma unroll
        for (int c = 0; c < FPL; ++c) dinv.v[c] = 1.0f / d.v[c];
        Vec<FPL> x, xold, r, z, p, ap;
#pragma unroll
        for (int c = 0; c < FPL; ++c) xold.v[c] = x.v[c] = xrow[f0 + c];  // warm start
        apply(x, ap);
        const float ynorm2 = vdot<FPL>(y, y);
#pragma unroll
        for (int c = 0; c < FPL; ++c) {
            r.v[c] = y.v[c] - ap.v[c];
            z.v[c] = r.v[c] * dinv.v[c];
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
                z.v[c] = r.v[c] * dinv.v[c];
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
        dd = cg_wave_sum(dd);
        if (lead) {
#pragma unroll
            for (int c = 0; c < FPL; ++c) xrow[f0 + c] = (f0 + c < k) ? x.v[c] : 0.f;
            if (lane == 0) row_delta[row] = dd;
        }
        if (__any(bad) && lead && lane == 0) atomicCAS(status, 0, row + 1);
        if (lead && lane == 0) {  // lk_als_plan_cg_stats: iterations and rows of this half-epoch
            atomicAdd(&status[2], it);
            atomicAdd(&status[3], 1);
        }
        if (ctl.d_done && lead && lane == 0) ctl_advance(ctl, 1);
        team_sync();
    }
}

static bool cg_wave_rows_enabled()
{
    const char *e = getenv("LK_ALS_CG_WAVE_ROWS");  // 0: every CG row by a workgroup (A-B knob)
    return !(e && e[0] == '0');
}

template <int KP, bool IS64>
static int launch_cg(const lk_als_plan *p, const void *indptr, const int32_t *indices,
                     const float *values, int64_t n_rows, int k, float *this_,
                     const float *other, const float *otor, int ld_otor, char *ws,
                     float *out_frob, hipStream_t st, int64_t t_begin)
{
    using IT = typename IndPtr<IS64>::type;
    int *status = reinterpret_cast<int *>(ws + p->off_status);
    float *row_delta = reinterpret_cast<float *>(ws + p->off_delta);
    float *partial = reinterpret_cast<float *>(ws + p->off_partial);
    if (t_begin == 0) LK_HIP_CHECK(hipMemsetAsync(status, 0, 64, st));  // (else: the exact stage did)
    if (p->ctl) {
        LK_HIP_CHECK(hipMemsetAsync(row_delta, 0, (size_t)n_rows * sizeof(float), st));
        int rc = ctl_begin(p->ctl, n_rows, n_rows, st);
        if (rc != LK_OK) return rc;
    }
    const int max_iter = p->cg_max_iter > 0 ? p->cg_max_iter : k;
    // rows that fit ONE wave's registers (<= 4096 / KP entries: tasks [t_cg1, n_rows)) take the
    // wave-per-row instance; not with a task-control block (the cancel poll is a workgroup matter)
    int64_t t_w1 = p->ctl || !cg_wave_rows_enabled() ? n_rows : p->t_cg1;
    if (t_w1 < t_begin) t_w1 = t_begin;
    const size_t lds = KP <= 128 ? (size_t)KP * KP * sizeof(float) : 0;  // OtOr resident
    auto k4 = als_cg_kernel<KP, IS64, 4>;
    auto k1 = als_cg_kernel<KP, IS64, 1>;
    if (lds > 0) {
        LK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k4),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        LK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k1),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (t_w1 > t_begin) {
        const int64_t n4 = t_w1 - t_begin;
        const int64_t blocks = n4 < 256 * 8 ? n4 : 256 * 8;
        hipLaunchKernelGGL(k4, dim3((unsigned)blocks), dim3(256), lds, st,
                           static_cast<const IT *>(indptr), indices, values, p->d_order, n_rows,
                           other, this_, otor, ld_otor, k, p->cg_tol, max_iter, row_delta,
                           status, p->ctl ? p->ctl->dev() : TaskCtlDev{}, t_begin, t_w1);
    }
    if (n_rows > t_w1) {
        const int64_t n1 = (n_rows - t_w1 + 3) / 4;
        const int64_t blocks = n1 < 256 * 8 ? n1 : 256 * 8;
        hipLaunchKernelGGL(k1, dim3((unsigned)blocks), dim3(256), lds, st,
                           static_cast<const IT *>(indptr), indices, values, p->d_order, n_rows,
                           other, this_, otor, ld_otor, k, p->cg_tol, max_iter, row_delta,
                           status, TaskCtlDev{}, t_w1, n_rows);
    }
    return launch_delta_reduce(row_delta, n_rows, partial, out_frob, st);
}

// LK_ALS_CG_HYBRID (test hook / A-B knob): 0 = CG for every row; 1 = the chunked rows (more than
// LK_ALS_LONG_ROW entries) go to the exact kernels; default 2 = every row longer than the CG
// kernel keeps in registers does.
static int cg_hybrid_mode()
{
    const char *e = getenv("LK_ALS_CG_HYBRID");
    return (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 2;
}

int als_cg_half_epoch(const lk_als_plan *p, const void *indptr, int is64, const int32_t *indices,
                      const float *values, int64_t n_rows, int k, float *this_, int ld_this,
                      const float *other, int ld_other, const float *otor, int ld_otor, char *ws,
                      float *out_frob, hipStream_t st)
{
    // Matrix-free CG pays for itself where a row's gathered factor rows stay in registers over
    // the iterations: 256 / 128 / 64 entries at padded k = 64 / 128 / 256 (t_cg: 85 % of the
    // ML-25M user rows, 96 % of its item rows, 40 % of the entries at k = 64).  Longer rows would
    // gather their tail again in EVERY iteration (measured at k = 64: 62 ms per epoch with CG on
    // every row -- one workgroup iterating over the 81 491-entry item is 2.5 ms per iteration of
    // serial gather latency; 29.6 ms with only the chunked rows solved exactly),
    // while the exact kernels form such a row's normal matrix once, on the matrix cores.  So the
    // first t_cg tasks of the longest-first order go to the exact kernels, the rest to CG.
    // (Not with a task-control block: its progress accounting is per launch.)
    int64_t t_begin = 0;
    const int mode = p->ctl ? 0 : cg_hybrid_mode();
    const int64_t t_exact = mode == 2 ? p->t_cg : (mode == 1 ? p->n_long : 0);
    if (t_exact > 0) {
        p->dense_limit = t_exact;
        const int rc =
            p->KP > 64 ? als_blk_half_epoch(p, indptr, is64, indices, values, n_rows, 0, k, this_,
                                            other, otor, ld_otor, ws, out_frob, st, false, 0.f)
                       : als_chol_half_epoch(p, indptr, is64, indices, values, n_rows, k, this_,
                                             ld_this, other, ld_other, otor, ld_otor, ws, out_frob,
                                             st);
        p->dense_limit = -1;
        if (rc != LK_OK) return rc;
        t_begin = t_exact;
    }
#define LK_CG_CASE(KPV)                                                                     \
    return is64 ? launch_cg<KPV, true>(p, indptr, indices, values, n_rows, k, this_, other, \
                                       otor, ld_otor, ws, out_frob, st, t_begin)            \
                : launch_cg<KPV, false>(p, indptr, indices, values, n_rows, k, this_, other, \
                                        otor, ld_otor, ws, out_frob, st, t_begin)
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

extern "C" int lk_als_plan_cg_stats(const lk_als_plan *plan, void *d_ws, void *stream,
                                    int64_t *out_iterations, int64_t *out_rows)
{
    LK_REQUIRE(plan && d_ws, "lk_als_plan_cg_stats: null pointer");
    int st[4] = {0, 0, 0, 0};
    LK_HIP_CHECK(hipMemcpyAsync(st, static_cast<char *>(d_ws) + plan->off_status, sizeof(st),
                                hipMemcpyDeviceToHost, lk::as_stream(stream)));
    LK_HIP_CHECK(hipStreamSynchronize(lk::as_stream(stream)));
    if (out_iterations) *out_iterations = st[2];
    if (out_rows) *out_rows = st[3];
    return LK_OK;
}
