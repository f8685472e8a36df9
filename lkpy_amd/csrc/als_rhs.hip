// als_rhs.hip -- the right-hand side of the ALS row solve IN THE REFERENCE'S ORDER.
//
// `train_row_solve` (src/accel/als/implicit.rs:116-117) forms  y = mt.dot(&vals)  with
// vals = v + 1 and mt = o_picked.t(), a TRANSPOSED view: its rows (one per feature) have stride k,
// so ndarray 0.17's mat-vec (`general_mat_vec_mul_impl`, no blas feature) takes `row.dot(x)` down
// the non-contiguous path of `dot_generic` -- a plain fold, one accumulator:
//
//     y[f] = 0;  for j in row order:  y[f] = round(y[f] + round(M[j][f] * (v_j + 1)))
//
// (Rust never contracts a*b+c into an FMA.)  Over a row of 10^4 .. 10^6 entries of one sign that
// single float32 chain drifts SYSTEMATICALLY: the busiest cfg5 item (1.54 M entries) lands 7e-2
// from the float64 sum, the busiest ML-25M item (81 491) 1.0e-4.  The
// solve kernels' own sums (four entry slots per wave, chunk slabs, slab groups) stay 1e-5 from
// float64 there -- closer to the truth, but not what the reference computes, and the north star
// asks for the reference's factors within 1e-4.
//
// This file evaluates that chain for the rows a plan names -- the rows longer than LK_ALS_REF_LEN
// entries of every default ("hybrid") plan, every dense row of a strict reference-order plan
// (als_plan.h) -- bit for bit (same order, same roundings; explicit.rs:110 is the same call with
// vals = the ratings), and the solve kernels take their right-hand side from it.
//
// Round 5: the chain is no longer latency bound.  A chain is 4 cycles per entry at best (one
// dependent v_add per entry); round 4's kernel (lane = feature, 16 gathers in flight) ran at the
// gather latency, ~75 cycles per entry: 50 ms for the 1.54 M-entry row.  Now one workgroup serves
// (row, 64-feature slice): wave 0 only ADDS -- it reads the products four entries at a time
// (`ds_read_b128`: the stage buffer is feature-major) and runs the chain, 5 instructions per 4
// entries; waves 1..RC_NP are producers: they gather the factor rows RC_D stages ahead into
// registers (coalesced 64-byte quads), multiply by (v + 1) -- the product is rounded exactly as
// in the reference, it just happens on another wave -- and write them transposed into the other
// half of a double buffer.  One barrier per stage of RC_E entries, no `s_waitcnt vmcnt(0)`
// anywhere in the loop (the barrier is an `s_barrier` preceded by `lgkmcnt(0)` only: hipcc's
// __syncthreads would drain the gather ring).
#include "als_plan.h"
#include "common.h"

#pragma clang fp contract(off)

namespace lk {

// Measured stand-alone on the ML-25M item half (longest row 81 491 entries; tools/chain_time.py,
// rocprofv3 kernel trace with LK_ALS_SIDE_STREAM=0): NP = 3, G = 2: 613 us = 16 cycles per entry,
// the producers slower than the chain; with packed multiplies and one select per entry 566 us;
// NP = 4: 451 ... 460 us.  In the epoch the two are equal (3.52 vs 3.60 ms at cfg2: what the
// chains take from the machine they take from the kernels they run beside), so the smaller
// workgroup is the default.  NP = 6, G = 1 would halve a producer's work per stage, but hipcc
// 7.2 then regroups the loads of a round (all q loads last) and waits for nearly all of them at
// the first use -- its s_waitcnt placement in this loop is only right for some shapes (checked in
// the ISA for the default: 24 .. 35 loads in flight at every wait); loads issued by inline asm
// with hand-counted waits are the way past that, not built.
#ifndef LK_RHS_NP
#define LK_RHS_NP 3  // producer waves per workgroup
#endif
#ifndef LK_RHS_G
#define LK_RHS_G 2  // 16-entry groups per producer wave and stage
#endif
#ifndef LK_RHS_D
#define LK_RHS_D (LK_RHS_G == 1 ? 4 : 3)  // (G = 2, D = 4: hipcc 7.2 runs out of registers and
                                          // parks the ring in AGPRs -- waits of 0)
#endif
constexpr int RC_NP = LK_RHS_NP;
constexpr int RC_G = LK_RHS_G;
constexpr int RC_E = RC_NP * RC_G * 16;  // entries per stage (a multiple of 32: the swizzle below)
constexpr int RC_D = LK_RHS_D;    // stages of gathered rows in flight per producer wave (registers)
static_assert(RC_E % 32 == 0 && RC_E % 4 == 0, "stage size");
constexpr int RC_F = 64;  // features per workgroup (one chain wave)

// stage buffer: product of (feature f, entry e) at word  f * RC_E + (e ^ ((f & 7) << 2)):
// a chain lane's ds_read_b128 of entries 4g .. 4g + 3 hits bank quad (g ^ f) & 7 -- eight
// consecutive lanes, eight quads, conflict free; the producers' transposed ds_write_b32 are
// 2-way conflicted (they have the slack)
__device__ __forceinline__ int rc_word(int f, int e) { return f * RC_E + (e ^ ((f & 7) << 2)); }

__device__ __forceinline__ void rc_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <bool IS64>
__global__ __launch_bounds__((RC_NP + 1) * 64) void als_rhs_chain_kernel(
    const typename IndPtr<IS64>::type *__restrict__ indptr, const int32_t *__restrict__ indices,
    const float *__restrict__ values, const int32_t *__restrict__ order, int64_t n_tasks,
    const float *__restrict__ other, int KP, int n_slices, int expl, float *__restrict__ y_out)
{
    __shared__ __attribute__((aligned(16))) float buf[2][RC_F * RC_E];
    const int64_t t = blockIdx.x / (unsigned)n_slices;
    const int sl = (int)(blockIdx.x - t * n_slices);
    if (t >= n_tasks) return;
    const int row = order ? order[t] : (int)t;
    const int64_t beg = indptr[row], end = indptr[row + 1];
    const int64_t n = end - beg;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = lane_id();
    const int fw = KP < RC_F ? KP : RC_F;  // features of this slice
    const int fbase = sl * RC_F;
    if (n <= 0) {  // (the solve kernels never read it: implicit.rs:98-101)
        if (wave == 0 && lane < fw) y_out[t * KP + fbase + lane] = 0.f;
        return;
    }
    // stages, rounded up to whole rounds of RC_D: the loops below have ONE shape (no remainder
    // path whose loads hipcc could sink or whose waits it would have to guess); the padding
    // stages hold +0.0 products, and y + 0.0 = y exactly
    const int nst = (int)((n + RC_E - 1) / RC_E + RC_D - 1) / RC_D * RC_D;

    if (wave == 0) {
        // ---- the chain: lane = feature ------------------------------------------------------
        __builtin_amdgcn_s_setprio(3);
        float y = 0.f;
        const int sw = (lane & 7) << 2;
        const float *mine = &buf[0][0] + lane * RC_E;
        auto consume = [&](int b) {
            const float *src = mine + b * (RC_F * RC_E);
            static_assert((RC_E / 4) % 8 == 0, "stage = whole batches of 8 reads");
#pragma unroll
            for (int g0 = 0; g0 < RC_E / 4; g0 += 8) {
                f32x4 r[8];
#pragma unroll
                for (int g = 0; g < 8; ++g)
                    r[g] = *reinterpret_cast<const f32x4 *>(src + ((((g0 + g) << 2) ^ sw)));
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    y = y + r[g].x;
                    y = y + r[g].y;
                    y = y + r[g].z;
                    y = y + r[g].w;
                }
            }
        };
        for (int s = 0; s < nst; ++s) {
            if (s > 0) consume((s - 1) & 1);
            rc_barrier();
        }
        consume((nst - 1) & 1);
        if (lane < fw) y_out[t * KP + fbase + lane] = y;
        return;
    }

    // ---- producers: wave pw takes the entry groups RC_G pw .. RC_G pw + RC_G - 1 (16 entries
    // each) of every stage; lane -> entry (lane >> 2) of the group, float4 column jg * 4 + (lane & 3)
    const int pw = wave - 1;
    const int el0 = (RC_G * pw) * 16 + (lane >> 2);  // group h of this wave: entry el0 + 16 h
    int foff[4];
#pragma unroll
    for (int jg = 0; jg < 4; ++jg) {
        const int f0 = (jg * 4 + (lane & 3)) * 4;
        foff[jg] = fbase + (f0 < fw ? f0 : 0);
    }
    const int32_t *ci = indices + beg;
    const float *cv = values + beg;
    const int64_t last = n - 1;

    f32x4 q[RC_D][RC_G][4];
    float v[RC_D][RC_G];
    int c[RC_D][RC_G];
    // (The arrays are `const __restrict__`: to hipcc their loads commute with every asm
    // statement, "memory" clobber or not, and it may regroup them across steps.  Laundering the
    // base pointers through "+s" asm operands pins them -- and makes hipcc 7.2 wait
    // vmcnt(0) lgkmcnt(0) in front of every such statement.  The configuration below is the one
    // whose generated waits were checked in the ISA: 24 .. 35 loads in flight at every wait.)
    auto load_idx = [&](int d, int s) {
        const int32_t *ci_ = ci;
#pragma unroll
        for (int h = 0; h < RC_G; ++h) {
            const int64_t e = (int64_t)s * RC_E + el0 + 16 * h;
            c[d][h] = ci_[e < last ? e : last];
        }
    };
    auto issue = [&](int d, int s) {
        const float *cv_ = cv, *oth = other;
#pragma unroll
        for (int h = 0; h < RC_G; ++h) {
            const int64_t e = (int64_t)s * RC_E + el0 + 16 * h;
            v[d][h] = cv_[e < last ? e : last];
        }
#pragma unroll
        for (int h = 0; h < RC_G; ++h) {
            const float *r = oth + (int64_t)c[d][h] * KP;
#pragma unroll
            for (int jg = 0; jg < 4; ++jg) q[d][h][jg] = *reinterpret_cast<const f32x4 *>(r + foff[jg]);
        }
    };
    auto write_out = [&](int d, int s) {
        float *dst = &buf[s & 1][0];
#pragma unroll
        for (int h = 0; h < RC_G; ++h) {
            const int el = el0 + 16 * h;
            const bool live = (int64_t)s * RC_E + el < n;
            // `vals += 1.0` (implicit.rs:116) rounded to f32; explicit.rs:110: the ratings.
            // Entries past the end multiply by +0.0: the chain then adds a zero, y + 0 = y exactly
            // (one select per entry instead of one per product)
            float v1 = expl ? v[d][h] : v[d][h] + 1.0f;
            v1 = live ? v1 : 0.f;
            // (pins this step's multiplications behind the previous step's barrier: volatile asm
            // statements keep their order, and without it hipcc hoists the products of EVERY ring
            // slot to the top of a round and waits vmcnt(0) there -- checked in the ISA)
            asm volatile("" : "+v"(v1));
            const f32x2 v2 = f32x2{v1, v1};
#pragma unroll
            for (int jg = 0; jg < 4; ++jg) {
                const int f0 = (jg * 4 + (lane & 3)) * 4;
                // rounded products, two per instruction (v_pk_mul_f32; never an FMA: nothing is
                // added here)
                const f32x2 lo = f32x2{q[d][h][jg].x, q[d][h][jg].y} * v2;
                const f32x2 hi = f32x2{q[d][h][jg].z, q[d][h][jg].w} * v2;
                dst[rc_word(f0 + 0, el)] = lo.x;
                dst[rc_word(f0 + 1, el)] = lo.y;
                dst[rc_word(f0 + 2, el)] = hi.x;
                dst[rc_word(f0 + 3, el)] = hi.y;
            }
        }
    };
    // The loop below must look the same to hipcc's s_waitcnt pass from both of its entries: the
    // prologue issues its loads in the order a loop step does (the empty asm statements keep the
    // scheduler from regrouping them) and every step is unconditional -- a stage's loads are then
    // waited for with `vmcnt(loads of the RC_D - 1 younger stages)`, never 0.
#pragma unroll
    for (int d = 0; d < RC_D; ++d) load_idx(d, d);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int d = 0; d < RC_D; ++d) {
        issue(d, d);
        load_idx(d, d + RC_D);
        asm volatile("" ::: "memory");
    }
    int s0 = 0;
    do {
#pragma unroll
        for (int d = 0; d < RC_D; ++d) {
            const int s = s0 + d;
            write_out(d, s);     // (stages past the row's end: zeros)
            issue(d, s + RC_D);  // (clamped to the row's last entry past the end)
            load_idx(d, s + 2 * RC_D);
            rc_barrier();
        }
        s0 += RC_D;
    } while (s0 < nst);
}

static bool rhs_side_enabled()
{
    const char *e = getenv("LK_ALS_SIDE_STREAM");
    return !(e && e[0] == '0');
}

int plan_fork_rhs(const lk_als_plan *p, hipStream_t st, hipStream_t *side)
{
    *side = st;
    if (!rhs_side_enabled()) return LK_OK;
    if (!p->side_rhs) {
        p->side_rhs = lk::side_stream_acquire();
        LK_REQUIRE(p->side_rhs != nullptr, "als: no side stream");
        LK_HIP_CHECK(hipEventCreateWithFlags(&p->ev_fork_rhs, hipEventDisableTiming));
        LK_HIP_CHECK(hipEventCreateWithFlags(&p->ev_join_rhs, hipEventDisableTiming));
        LK_HIP_CHECK(hipEventCreateWithFlags(&p->ev_mid_rhs, hipEventDisableTiming));
    }
    LK_HIP_CHECK(hipEventRecord(p->ev_fork_rhs, st));
    LK_HIP_CHECK(hipStreamWaitEvent(p->side_rhs, p->ev_fork_rhs, 0));
    *side = p->side_rhs;
    return LK_OK;
}

int plan_rhs_wait_main(const lk_als_plan *p, hipStream_t st)
{
    if (!rhs_side_enabled() || !p->side_rhs) return LK_OK;
    LK_HIP_CHECK(hipEventRecord(p->ev_mid_rhs, st));
    LK_HIP_CHECK(hipStreamWaitEvent(p->side_rhs, p->ev_mid_rhs, 0));
    return LK_OK;
}

int plan_join_rhs(const lk_als_plan *p, hipStream_t st)
{
    if (!rhs_side_enabled() || !p->side_rhs) return LK_OK;
    LK_HIP_CHECK(hipEventRecord(p->ev_join_rhs, p->side_rhs));
    LK_HIP_CHECK(hipStreamWaitEvent(st, p->ev_join_rhs, 0));
    return LK_OK;
}

// tasks [0, n_tasks) of `order` (order == nullptr: rows 0 .. n_tasks); y_out[t] <- row order[t]
int launch_rhs_reference(const lk_als_plan *p, const void *indptr, int is64,
                         const int32_t *indices, const float *values, const int32_t *order,
                         int64_t n_tasks, const float *other, bool expl, float *y_out,
                         hipStream_t st)
{
    if (!y_out || n_tasks <= 0) return LK_OK;
    const int n_slices = (p->KP + RC_F - 1) / RC_F;
    LK_REQUIRE(n_tasks * n_slices < (int64_t)INT32_MAX, "rhs chain: grid too large");
    const dim3 grid((unsigned)(n_tasks * n_slices)), block((RC_NP + 1) * 64);
    if (is64)
        hipLaunchKernelGGL(als_rhs_chain_kernel<true>, grid, block, 0, st,
                           static_cast<const int64_t *>(indptr), indices, values, order, n_tasks,
                           other, p->KP, n_slices, expl ? 1 : 0, y_out);
    else
        hipLaunchKernelGGL(als_rhs_chain_kernel<false>, grid, block, 0, st,
                           static_cast<const int32_t *>(indptr), indices, values, order, n_tasks,
                           other, p->KP, n_slices, expl ? 1 : 0, y_out);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

}  // namespace lk

extern "C" int lk_als_plan_set_rhs_workspace(lk_als_plan *p, float *d_y)
{
    LK_REQUIRE(p != nullptr, "lk_als_plan_set_rhs_workspace: null plan");
    LK_REQUIRE(d_y == nullptr || p->solver == LK_SOLVER_CHOLESKY,
               "lk_als_plan_set_rhs_workspace: the reference-order right-hand side belongs to the "
               "exact solver (the CG option has no reference to reproduce)");
    p->d_yref = d_y;
    return LK_OK;
}
