// als_rhs.hip -- the right-hand side of the ALS row solve IN THE REFERENCE'S ORDER (optional).
//
// `train_row_solve` (src/accel/als/implicit.rs:116-117) forms  y = mt.dot(&vals)  with
// vals = v + 1 and mt = o_picked.t(), a TRANSPOSED view: its rows (one per feature) have stride k,
// so ndarray 0.17's mat-vec (`general_mat_vec_mul_impl`, no blas feature) takes `row.dot(x)` down
// the non-contiguous path of `dot_generic` -- a plain fold, one accumulator:
//
//     y[f] = 0;  for j in row order:  y[f] = round(y[f] + round(M[j][f] * (v_j + 1)))
//
// (Rust never contracts a*b+c into an FMA.)  Over a row of 10^5 .. 10^6 entries of one sign that
// single float32 chain stagnates: the busiest cfg5 item (1.54 M entries) lands 7e-2 from the
// float64 sum, the busiest ML-25M item (81 491) 1.0e-4.  The solve kernels
// sum y pairwise-ish (four entry slots per wave, chunk slabs, slab groups) and are 1e-5 from
// float64 there -- closer to the truth, but not what the reference computes.
//
// With a rhs workspace attached to the plan (`lk_als_plan_set_rhs_workspace`; Python:
// LK_ALS_RHS_ORDER=reference) every half-epoch first runs THIS kernel -- lane = feature, the row's
// entries strictly in order, product and sum rounded separately -- and the solve kernels take
// their right-hand side from it instead of from their own accumulation.  The normal matrix and
// the factorisation are unchanged.  It reproduces the reference's y bit for bit (same order,
// same roundings: explicit.rs:110 is the same call with vals = the ratings), so the rows where
// the default mode is ">1e-4 from the oracle because the ORACLE drifts" come out within 1e-4 of
// it (tests/test_gpu_als_rhs_order.py).  A diagnostic / strict-reproduction mode: one lane chain
// per feature is latency bound (the 1.54 M-entry row alone takes ~50 ms).  Rows that go through
// the Woodbury kernels (<= 64 entries at padded k > 64) never form y and are not affected.
#include "als_plan.h"
#include "common.h"

#pragma clang fp contract(off)

namespace lk {

constexpr int RHS_BATCH = 16;  // gathered values in flight per lane

// One workgroup of max(KP, 64) threads per task: thread f owns feature f of row order[t].
template <bool IS64>
__global__ void als_rhs_reference_kernel(const typename IndPtr<IS64>::type *__restrict__ indptr,
                                         const int32_t *__restrict__ indices,
                                         const float *__restrict__ values,
                                         const int32_t *__restrict__ order, int64_t n_tasks,
                                         const float *__restrict__ other, int KP, int expl,
                                         float *__restrict__ y_out)
{
    const int64_t t = blockIdx.x;
    if (t >= n_tasks) return;
    const int row = order ? order[t] : (int)t;
    const int f = threadIdx.x;
    const int64_t beg = indptr[row], end = indptr[row + 1];
    const bool act = f < KP;
    const float *col = other + (act ? f : 0);
    float y = 0.f;
    for (int64_t b = beg; b < end; b += RHS_BATCH) {
        const int nb = (end - b) < RHS_BATCH ? (int)(end - b) : RHS_BATCH;
        float q[RHS_BATCH], v1[RHS_BATCH];
        // every load unconditional (entries past the end re-read the last one and are not summed)
#pragma unroll
        for (int j = 0; j < RHS_BATCH; ++j) {
            const int64_t e = j < nb ? b + j : end - 1;
            const int c = indices[e];  // wave-uniform: scalar loads
            const float v = values[e];
            q[j] = col[(int64_t)c * KP];
            v1[j] = expl ? v : v + 1.0f;  // `vals += 1.0` (implicit.rs:116), rounded to f32
        }
#pragma unroll
        for (int j = 0; j < RHS_BATCH; ++j) {
            if (j < nb) {
                float prod = q[j] * v1[j];
                asm volatile("" : "+v"(prod));  // keep hipcc from fusing the pair into v_fmac
                y = y + prod;
            }
        }
    }
    if (act) y_out[(int64_t)row * KP + f] = y;
}

// rows order[0 .. n_tasks) of the plan (order == nullptr: rows 0 .. n_tasks)
int launch_rhs_reference(const lk_als_plan *p, const void *indptr, int is64,
                         const int32_t *indices, const float *values, const int32_t *order,
                         int64_t n_tasks, const float *other, bool expl, hipStream_t st)
{
    if (!p->d_yref || n_tasks <= 0) return LK_OK;
    const dim3 grid((unsigned)n_tasks), block((unsigned)(p->KP < 64 ? 64 : p->KP));
    if (is64)
        hipLaunchKernelGGL(als_rhs_reference_kernel<true>, grid, block, 0, st,
                           static_cast<const int64_t *>(indptr), indices, values, order, n_tasks,
                           other, p->KP, expl ? 1 : 0, p->d_yref);
    else
        hipLaunchKernelGGL(als_rhs_reference_kernel<false>, grid, block, 0, st,
                           static_cast<const int32_t *>(indptr), indices, values, order, n_tasks,
                           other, p->KP, expl ? 1 : 0, p->d_yref);
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
