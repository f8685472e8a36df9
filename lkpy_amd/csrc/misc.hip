// misc.hip -- error plumbing, device query, pad/unpad of factor matrices.
#include <stdarg.h>

#include "common.h"

namespace lk {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// [n x k] (ld_src) -> [n x ld_dst], zero pad columns
__global__ void pad_rows_kernel(const float *__restrict__ src, int64_t n, int k, int ld_src,
                                float *__restrict__ dst, int ld_dst)
{
    const int64_t total = n * ld_dst;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ld_dst;
        const int c = (int)(i - r * ld_dst);
        dst[i] = (c < k) ? src[r * ld_src + c] : 0.f;
    }
}

__global__ void unpad_rows_kernel(const float *__restrict__ src, int64_t n, int k, int ld_src,
                                  float *__restrict__ dst, int ld_dst)
{
    const int64_t total = n * k;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / k;
        const int c = (int)(i - r * k);
        dst[r * ld_dst + c] = src[r * ld_src + c];
    }
}

static unsigned grid_for(int64_t total)
{
    int64_t b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace lk

extern "C" const char *lk_last_error(void) { return lk::g_err; }
extern "C" const char *lk_version(void) { return "lkpy_amd 0.1.0 (gfx950)"; }

extern "C" int lk_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int lk_pad_rows(const float *d_src, int64_t n, int32_t k, int32_t ld_src, float *d_dst,
                           int32_t ld_dst, void *stream)
{
    LK_REQUIRE(n >= 0 && k >= 1 && ld_src >= k && ld_dst >= k, "lk_pad_rows: bad shape");
    if (n == 0) return LK_OK;
    LK_REQUIRE(d_src && d_dst, "lk_pad_rows: null pointer");
    hipLaunchKernelGGL(lk::pad_rows_kernel, dim3(lk::grid_for(n * ld_dst)), dim3(256), 0,
                       lk::as_stream(stream), d_src, n, k, ld_src, d_dst, ld_dst);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}

extern "C" int lk_unpad_rows(const float *d_src, int64_t n, int32_t k, int32_t ld_src,
                             float *d_dst, int32_t ld_dst, void *stream)
{
    LK_REQUIRE(n >= 0 && k >= 1 && ld_src >= k && ld_dst >= k, "lk_unpad_rows: bad shape");
    if (n == 0) return LK_OK;
    LK_REQUIRE(d_src && d_dst, "lk_unpad_rows: null pointer");
    hipLaunchKernelGGL(lk::unpad_rows_kernel, dim3(lk::grid_for(n * k)), dim3(256), 0,
                       lk::as_stream(stream), d_src, n, k, ld_src, d_dst, ld_dst);
    LK_HIP_CHECK(hipGetLastError());
    return LK_OK;
}
