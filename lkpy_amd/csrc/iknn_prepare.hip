// iknn_prepare.hip -- item-kNN rating normalisation on the device, gfx950.
//
// `ItemKNNScorer.train` centres every item's ratings on the item mean (explicit feedback
// only) and scales every item vector to unit L2 norm before the similarity build
// (`_center_ratings` / `_normalize_rows`, src/lenskit/knn/item.py:202-228).  The reference
// does it with SciPy; what that arithmetic is, operation by operation (SciPy 1.15):
//
//   sums    = np.add.reduceat over each item's ratings in ascending-user order   (f32)
//   means   = f32(f64(sum) / f64(count))
//   c       = r - mean[item]                                                      (f32)
//   sumsq   = for every item, |c|^2 added ONE BY ONE in ascending-user order      (f32;
//             spla.norm converts to CSR and sums axis 0 through a csc_matvec loop)
//   norm    = sqrt(sumsq);  recip = 1 / max(norm, FLT_MIN);  value = c * recip[item]
//
// Everything elementwise is a single correctly rounded f32 operation and is done here; the
// sequential sum of squares is done here in the same order; the two tiny per-item vectors
// that need the host's libm / SIMD summation order (reduceat sums, sqrt / reciprocal) are
// computed by the caller with the very NumPy calls the reference makes.  The result is
// bit-identical to the reference's preparation (tests/test_gpu_iknn_prepare.py).
//
// Input is the item-major (transposed) rating matrix from lk_csr_transpose together with
// its permutation, so the user-major values are produced by a scatter.
#include "common.h"

// sequential f32 sums: no FMA contraction anywhere in this file
#pragma clang fp contract(off)

namespace lk {

// One wave per item: lanes load / centre / square 64 entries at a time (coalesced), lane
// order = entry order; the running sum is carried sequentially through the 64 squares by
// a lane-ordered chain (readlane of each square in turn), i.e. exactly
//   for e in entries: acc = acc + sq[e].
template <bool IS64>
__global__ __launch_bounds__(256) void iknn_prep_center_kernel(
    const typename IndPtr<IS64>::type *__restrict__ t_ptr, const float *__restrict__ vals,
    const float *__restrict__ means, int64_t n_items, float *__restrict__ cent,
    float *__restrict__ sumsq, int *__restrict__ nonzero_flag)
{
    const int lane = lane_id();
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n_items) return;
    const int64_t b = t_ptr[item], e = t_ptr[item + 1];
    const float mu = means ? means[item] : 0.f;
    float acc = 0.f;
    bool any = false;
    for (int64_t base = b; base < e; base += 64) {
        const int64_t i = base + lane;
        float sq = 0.f;
        if (i < e) {
            float c = vals[i];
            if (means) c = c - mu;  // rmat.data - np.repeat(means, counts)
            cent[i] = c;
            const float a = fabsf(c);
            sq = a * a;  // abs(x).power(2)
            any = any || (a > 1.0e-8f);  // np.allclose(data, 0): |x| <= 1e-8
        }
        const int n = (int)((e - base) < 64 ? (e - base) : 64);
        for (int l = 0; l < n; ++l) {
            const float s = bcast(sq, l);
            acc = acc + s;
            asm volatile("" : "+v"(acc));  // keep one rounded add per entry
        }
    }
    if (lane == 0) sumsq[item] = acc;
    if (__any(any) && lane == 0) atomicOr(nonzero_flag, 1);
}

// value = c * recip[item] in item-major order, scattered to user-major order through perm
template <bool IS64>
__global__ __launch_bounds__(256) void iknn_prep_scale_kernel(
    const typename IndPtr<IS64>::type *__restrict__ t_ptr,
    const typename IndPtr<IS64>::type *__restrict__ perm, const float *__restrict__ cent,
    const float *__restrict__ recip, int64_t n_items, float *__restrict__ val_items,
    float *__restrict__ val_users)
{
    const int lane = lane_id();
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n_items) return;
    const int64_t b = t_ptr[item], e = t_ptr[item + 1];
    const float r = recip[item];
    for (int64_t i = b + lane; i < e; i += 64) {
        const float v = cent[i] * r;
        val_items[i] = v;
        val_users[perm[i]] = v;
    }
}

}  // namespace lk

extern "C" int lk_iknn_prep_center(const void *d_item_indptr, int indptr_is_64,
                                   const float *d_item_values, const float *d_means,
                                   int64_t n_items, float *d_centered, float *d_sumsq,
                                   int32_t *d_nonzero_flag, void *stream)
{
    LK_REQUIRE(n_items >= 0, "lk_iknn_prep_center: negative size");
    LK_REQUIRE(d_item_indptr && d_sumsq && d_nonzero_flag, "lk_iknn_prep_center: null pointer");
    hipStream_t st = lk::as_stream(stream);
    LK_HIP_CHECK(hipMemsetAsync(d_nonzero_flag, 0, sizeof(int32_t), st));
    if (n_items == 0) return LK_OK;
    const unsigned grid = (unsigned)((n_items + 3) / 4);
    if (indptr_is_64)
        hipLaunchKernelGGL(lk::iknn_prep_center_kernel<true>, dim3(grid), dim3(256), 0, st,
                           static_cast<const int64_t *>(d_item_indptr), d_item_values, d_means,
                           n_items, d_centered, d_sumsq, d_nonzero_flag);
    else
        hipLaunchKernelGGL(lk::iknn_prep_center_kernel<false>, dim3(grid), dim3(256), 0, st,
                           static_cast<const int32_t *>(d_item_indptr), d_item_values, d_means,
                           n_items, d_centered, d_sumsq, d_nonzero_flag);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

extern "C" int lk_iknn_prep_scale(const void *d_item_indptr, int indptr_is_64, const void *d_perm,
                                  const float *d_centered, const float *d_recip, int64_t n_items,
                                  float *d_item_values_out, float *d_user_values_out,
                                  void *stream)
{
    LK_REQUIRE(n_items >= 0, "lk_iknn_prep_scale: negative size");
    LK_REQUIRE(d_item_indptr && d_recip, "lk_iknn_prep_scale: null pointer");
    if (n_items == 0) return LK_OK;
    hipStream_t st = lk::as_stream(stream);
    const unsigned grid = (unsigned)((n_items + 3) / 4);
    if (indptr_is_64)
        hipLaunchKernelGGL(lk::iknn_prep_scale_kernel<true>, dim3(grid), dim3(256), 0, st,
                           static_cast<const int64_t *>(d_item_indptr),
                           static_cast<const int64_t *>(d_perm), d_centered, d_recip, n_items,
                           d_item_values_out, d_user_values_out);
    else
        hipLaunchKernelGGL(lk::iknn_prep_scale_kernel<false>, dim3(grid), dim3(256), 0, st,
                           static_cast<const int32_t *>(d_item_indptr),
                           static_cast<const int32_t *>(d_perm), d_centered, d_recip, n_items,
                           d_item_values_out, d_user_values_out);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}
