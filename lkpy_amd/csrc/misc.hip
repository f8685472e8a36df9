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

// ---------------------------------------------------------------------------
// lk_task_ctl: cancel + progress words (AccelTask.cancel / current_progress,
// src/accel/tasks/mod.rs:62-106, src/lenskit/parallel/_task.py:34-57)
// ---------------------------------------------------------------------------

namespace lk {

int ctl_begin(lk_task_ctl *ctl, int64_t rows_total, int64_t units_total, hipStream_t st)
{
    ctl->rows_total = rows_total;
    ctl->units_total = units_total;
    // progress restarts; a cancel requested before the launch stays visible (h_words[0])
    *reinterpret_cast<volatile unsigned long long *>(ctl->h_words + 2) = 0ull;
    LK_HIP_CHECK(hipMemsetAsync(ctl->d_words, 0, 16, st));
    // a cancel requested before the launch is planted in HBM directly: no row starts at all
    // (otherwise every resident workgroup would begin before the first PCIe poll returns)
    if (__atomic_load_n(ctl->h_words, __ATOMIC_ACQUIRE))
        LK_HIP_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ctl->d_words), 1, 1, st));
    return LK_OK;
}

int ctl_finish(lk_task_ctl *ctl, hipStream_t st)
{
    int w[4] = {0, 0, 0, 0};
    LK_HIP_CHECK(hipMemcpyAsync(w, ctl->d_words, sizeof(w), hipMemcpyDeviceToHost, st));
    LK_HIP_CHECK(hipStreamSynchronize(st));
    unsigned long long done;
    memcpy(&done, w + 2, 8);
    *reinterpret_cast<volatile unsigned long long *>(ctl->h_words + 2) = done;
    if (w[0] != 0) {
        set_error("cancelled after %llu of %lld work units", done, (long long)ctl->units_total);
        return LK_E_CANCELLED;
    }
    return LK_OK;
}

}  // namespace lk

extern "C" int lk_task_ctl_create(lk_task_ctl **out)
{
    LK_REQUIRE(out != nullptr, "lk_task_ctl_create: null pointer");
    auto *c = new lk_task_ctl();
    hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&c->h_words), 64,
                                 hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) {
        memset(c->h_words, 0, 64);
        e = hipHostGetDevicePointer(reinterpret_cast<void **>(&c->dh_words), c->h_words, 0);
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->d_words), 64);
    if (e == hipSuccess) e = hipMemset(c->d_words, 0, 64);
    if (e != hipSuccess) {
        lk::set_error("lk_task_ctl_create: %s", hipGetErrorString(e));
        lk_task_ctl_destroy(c);
        return LK_E_HIP;
    }
    *out = c;
    return LK_OK;
}

extern "C" void lk_task_ctl_destroy(lk_task_ctl *c)
{
    if (!c) return;
    if (c->d_words) (void)hipFree(c->d_words);
    if (c->h_words) (void)hipHostFree(c->h_words);
    delete c;
}

extern "C" void lk_task_ctl_cancel(lk_task_ctl *c)
{
    if (c && c->h_words) __atomic_store_n(c->h_words, 1, __ATOMIC_RELEASE);
}

extern "C" int lk_task_ctl_cancelled(const lk_task_ctl *c)
{
    return (c && c->h_words) ? __atomic_load_n(c->h_words, __ATOMIC_ACQUIRE) : 0;
}

extern "C" void lk_task_ctl_reset(lk_task_ctl *c)
{
    if (!c || !c->h_words) return;
    __atomic_store_n(c->h_words, 0, __ATOMIC_RELEASE);
    *reinterpret_cast<volatile unsigned long long *>(c->h_words + 2) = 0ull;
}

extern "C" int lk_task_ctl_progress(const lk_task_ctl *c, int64_t *rows_done, int64_t *rows_total)
{
    LK_REQUIRE(c && rows_done && rows_total, "lk_task_ctl_progress: null pointer");
    const unsigned long long done =
        *reinterpret_cast<const volatile unsigned long long *>(c->h_words + 2);
    *rows_total = c->rows_total;
    *rows_done = c->units_total > 0
                     ? (int64_t)((__int128)done * c->rows_total / c->units_total)
                     : 0;
    return LK_OK;
}
