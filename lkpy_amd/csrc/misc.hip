// misc.hip -- error plumbing, device query, pad/unpad of factor matrices.
#include <stdarg.h>

#include <mutex>
#include <utility>
#include <vector>

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

namespace {
struct SideStreamPool {
    std::mutex mu;
    std::vector<std::pair<int, hipStream_t>> idle;  // (device, stream)
};
SideStreamPool &side_pool()
{
    static SideStreamPool *pool = new SideStreamPool();  // (never destroyed: outlives every plan)
    return *pool;
}
}  // namespace

hipStream_t side_stream_acquire()
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    {
        SideStreamPool &pl = side_pool();
        std::lock_guard<std::mutex> lock(pl.mu);
        for (size_t i = 0; i < pl.idle.size(); ++i)
            if (pl.idle[i].first == dev) {
                hipStream_t s = pl.idle[i].second;
                pl.idle.erase(pl.idle.begin() + (long)i);
                return s;
            }
    }
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
    return s;
}

void side_stream_release(hipStream_t s)
{
    if (!s) return;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipStreamDestroy(s);
        return;
    }
    SideStreamPool &pl = side_pool();
    std::lock_guard<std::mutex> lock(pl.mu);
    if (pl.idle.size() >= 64) {
        (void)hipStreamDestroy(s);
        return;
    }
    pl.idle.emplace_back(dev, s);
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

// ---------------------------------------------------------------------------------------------
// lk_download: device -> PAGEABLE host memory at PCIe speed.
//
// The reference hands its results to Python as host Arrow arrays (e.g. the similarity matrix of
// `compute_similarities`, src/accel/knn/item_train.rs:86-91; SURVEY.md section 8d counts the kNN
// build "to CSR sim matrix on host").  An unbounded ML-25M model is 9.2 GB: hipMemcpy into a
// fresh pageable buffer runs at ~12 GB/s -- one thread paying the page faults of 2.3 M fresh
// pages and the copy out of the driver's bounce buffer.  Here the transfer is chunked through a
// small ring of PINNED staging slots (DMA at link speed, hipMemcpyAsync) and a team of host
// threads copies the landed chunks to their final place in parallel, first-touching the
// destination pages on many cores at once.
// ---------------------------------------------------------------------------------------------
#include <atomic>
#include <mutex>
#include <thread>
#include <vector>
#include <string.h>

namespace lk {
namespace {
// 24 slots of 8 MiB: 192 MiB in flight is 3.7 ms of link time -- ample -- and pinning the ring
// (a one-time cost of the first large download of a process) stays ~20 ms; the rounds 1-2 ring of
// 16 x 16 MiB cost ~50 ms to pin and fed 8 host threads, which were the bottleneck (32 GB/s;
// 16 threads reach the link's 52 GB/s, tools/download_bench.py)
constexpr size_t DL_CHUNK = (size_t)8 << 20;  // bytes per staging slot
constexpr int DL_SLOTS = 24;

struct DownloadRing {
    char *slot[DL_SLOTS] = {};
    hipEvent_t done[DL_SLOTS] = {};
    hipStream_t stream = nullptr;
    bool ok = false;
    std::mutex mu;
    int init()
    {
        if (ok) return LK_OK;
        for (int i = 0; i < DL_SLOTS; ++i) {
            LK_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&slot[i]), DL_CHUNK, hipHostMallocDefault));
            LK_HIP_CHECK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
        }
        LK_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        ok = true;
        return LK_OK;
    }
};
DownloadRing g_ring;  // one transfer at a time (guarded by its mutex); slots stay pinned
}  // namespace
}  // namespace lk

namespace lk {
// WIDEN = 0: plain bytes.  WIDEN = 1: the source is uint16, the destination int32 (twice the
// bytes): the host team widens while it copies out of the staging slot, so half the bytes cross
// PCIe (lk_download_u16_as_i32: column indices of a matrix with <= 65 536 columns).
template <int WIDEN>
static int download_impl(void *h_dst, const void *d_src, size_t bytes, int32_t n_threads,
                         void *stream)
{
    if (bytes == 0) return LK_OK;
    // the producer of d_src ran on `stream`
    LK_HIP_CHECK(hipStreamSynchronize(lk::as_stream(stream)));
    char *dst = static_cast<char *>(h_dst);
    const char *src = static_cast<const char *>(d_src);
    auto put = [&](size_t off, const char *from, size_t n) {
        if (WIDEN) {
            const uint16_t *a = reinterpret_cast<const uint16_t *>(from);
            int32_t *b = reinterpret_cast<int32_t *>(dst + 2 * off);
            const size_t cnt = n / 2;
            for (size_t i = 0; i < cnt; ++i) b[i] = (int32_t)a[i];
        } else {
            memcpy(dst + off, from, n);
        }
    };
    if (bytes < 4 * lk::DL_CHUNK) {  // small: a plain copy
        if (!WIDEN) {
            LK_HIP_CHECK(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
            return LK_OK;
        }
        std::vector<char> tmp(bytes);
        LK_HIP_CHECK(hipMemcpy(tmp.data(), d_src, bytes, hipMemcpyDeviceToHost));
        put(0, tmp.data(), bytes);
        return LK_OK;
    }
    std::lock_guard<std::mutex> guard(lk::g_ring.mu);
    int rc = lk::g_ring.init();
    if (rc != LK_OK) return rc;
    if (n_threads < 1) n_threads = 16;
    if (n_threads > lk::DL_SLOTS - 2) n_threads = lk::DL_SLOTS - 2;
    const size_t n_chunks = (bytes + lk::DL_CHUNK - 1) / lk::DL_CHUNK;
    // chunk c lives in slot c % DL_SLOTS; issued / copied[c] hand the slots over
    std::atomic<size_t> issued{0};
    std::vector<std::atomic<int>> copied(n_chunks);
    for (auto &x : copied) x.store(0, std::memory_order_relaxed);
    std::atomic<int> failed{0};

    auto worker = [&](int t) {
        for (size_t c = (size_t)t; c < n_chunks; c += (size_t)n_threads) {
            while (issued.load(std::memory_order_acquire) <= c) {
                if (failed.load()) return;
                std::this_thread::yield();
            }
            const int s = (int)(c % lk::DL_SLOTS);
            if (hipEventSynchronize(lk::g_ring.done[s]) != hipSuccess) {
                failed.store(1);
                return;
            }
            const size_t off = c * lk::DL_CHUNK;
            const size_t n = bytes - off < lk::DL_CHUNK ? bytes - off : lk::DL_CHUNK;
            put(off, lk::g_ring.slot[s], n);
            copied[c].store(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> team;
    team.reserve((size_t)n_threads);
    for (int t = 0; t < n_threads; ++t) team.emplace_back(worker, t);
    hipError_t err = hipSuccess;
    for (size_t c = 0; c < n_chunks && !failed.load(); ++c) {
        if (c >= (size_t)lk::DL_SLOTS) {  // the slot's previous chunk must have been copied out
            const size_t prev = c - lk::DL_SLOTS;
            while (!copied[prev].load(std::memory_order_acquire)) {
                if (failed.load()) break;
                std::this_thread::yield();
            }
        }
        const int s = (int)(c % lk::DL_SLOTS);
        const size_t off = c * lk::DL_CHUNK;
        const size_t n = bytes - off < lk::DL_CHUNK ? bytes - off : lk::DL_CHUNK;
        err = hipMemcpyAsync(lk::g_ring.slot[s], src + off, n, hipMemcpyDeviceToHost,
                             lk::g_ring.stream);
        if (err == hipSuccess) err = hipEventRecord(lk::g_ring.done[s], lk::g_ring.stream);
        if (err != hipSuccess) {
            failed.store(1);
            break;
        }
        issued.store(c + 1, std::memory_order_release);
    }
    for (auto &th : team) th.join();
    if (failed.load()) {
        lk::set_error("lk_download: transfer failed: %s", hipGetErrorString(err));
        return LK_E_HIP;
    }
    return LK_OK;
}

__global__ void narrow_i32_u16_kernel(const int32_t *__restrict__ src, int64_t n,
                                      uint16_t *__restrict__ dst)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = (uint16_t)src[i];
}
}  // namespace lk

// Pin the staging ring now (idempotent): lets a caller keep the one-time cost out of a timed call.
extern "C" int lk_download_warmup(void)
{
    std::lock_guard<std::mutex> guard(lk::g_ring.mu);
    return lk::g_ring.init();
}

extern "C" int lk_download(void *h_dst, const void *d_src, size_t bytes, int32_t n_threads,
                           void *stream)
{
    LK_REQUIRE(bytes == 0 || (h_dst && d_src), "lk_download: null pointer");
    return lk::download_impl<0>(h_dst, d_src, bytes, n_threads, stream);
}

extern "C" int lk_download_i32_narrow(int32_t *h_dst, const int32_t *d_src, int64_t n,
                                      void *d_tmp_u16, int32_t n_threads, void *stream)
{
    LK_REQUIRE(n >= 0 && (n == 0 || (h_dst && d_src && d_tmp_u16)),
               "lk_download_i32_narrow: null pointer");
    if (n == 0) return LK_OK;
    hipLaunchKernelGGL(lk::narrow_i32_u16_kernel, dim3(4096), dim3(256), 0, lk::as_stream(stream),
                       d_src, n, static_cast<uint16_t *>(d_tmp_u16));
    LK_HIP_CHECK(hipGetLastError());
    return lk::download_impl<1>(h_dst, d_tmp_u16, (size_t)n * 2, n_threads, stream);
}
