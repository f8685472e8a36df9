// common.h -- shared helpers for the gfx950 kernels and the C-ABI glue.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>

#include "../../include/lkamd.h"

namespace lk {

// thread-local last-error message behind lk_last_error()
void set_error(const char *fmt, ...);

#define LK_HIP_CHECK(expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            lk::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,   \
                          __LINE__);                                                         \
            return LK_E_HIP;                                                                 \
        }                                                                                    \
    } while (0)

#define LK_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            lk::set_error(__VA_ARGS__);  \
            return LK_E_INVALID;         \
        }                                \
    } while (0)

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Non-blocking side streams from a per-device pool (misc.hip): a plan that lives for one call (the
// fold-in of a batch of queries) would otherwise create and destroy two streams per call -- a
// quarter of a millisecond.  A released stream may still hold queued work: whoever takes it next
// simply queues behind it.
hipStream_t side_stream_acquire();
void side_stream_release(hipStream_t s);

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int WAVE = 64;

// CSR offsets are int32 (Arrow List) or int64 (LargeList); kernels are
// instantiated for both.
template <bool IS64>
struct IndPtr;
template <>
struct IndPtr<false> {
    using type = int32_t;
};
template <>
struct IndPtr<true> {
    using type = int64_t;
};

// ---- wave-level primitives (wave64) --------------------------------------

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// Lanes of ONE wave exchanging data through LDS: the hardware keeps a wave's LDS operations in
// order, but to the compiler a store by one lane and a load of the same address by another are
// unrelated -- it may move the load above the store (seen in als_wb4_kernel: stale right-hand
// sides).  Wavefront-scope release / acquire fences around a wave barrier cost no instruction
// and pin the order.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// broadcast lane `src` (compile-time or wave-uniform) of v
__device__ __forceinline__ float bcast(float v, int src)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}

// ---- order-preserving float <-> unsigned key (top-N selection / sorting) -------------------
__device__ __forceinline__ unsigned f2key(float x)
{
    unsigned u = __builtin_bit_cast(unsigned, x);
    if (u == 0x80000000u) u = 0u;  // -0.0 ranks with +0.0 (they compare equal)
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k)
{
    const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __builtin_bit_cast(float, u);
}

// ---- cooperative cancel + live progress (lk_task_ctl, include/lkamd.h) -------------------
// Device view of a task-control block, passed BY VALUE to the long-running kernels; all
// null = no control block (the kernels are instantiated without the checks for that case).
struct TaskCtlDev {
    const int *h_cancel = nullptr;          // pinned, device-mapped host word: host sets it to 1
    unsigned long long *h_done = nullptr;   // pinned host word: live count of finished units
    int *d_cancel = nullptr;                // HBM: 1 once any workgroup has seen the cancel
    unsigned long long *d_done = nullptr;   // HBM: exact count of finished units
};

// One lane per workgroup/wave calls this before starting a unit.  The pinned host word is
// polled over PCIe only when `poll_host` (a sparse subset of workgroups); everybody else
// reads the HBM flag those pollers raise -- one L2-served load.
__device__ __forceinline__ bool ctl_cancelled(const TaskCtlDev &c, bool poll_host)
{
    int f = __hip_atomic_load(c.d_cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!f && poll_host) {
        f = __hip_atomic_load(c.h_cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (f) __hip_atomic_store(c.d_cancel, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return f != 0;
}

// `n` units finished (one lane calls it).  Every 256th unit the running total is also
// posted to the pinned host word, where lk_task_ctl_progress reads it without touching the
// device.
__device__ __forceinline__ void ctl_advance(const TaskCtlDev &c, unsigned n)
{
    const unsigned long long old = atomicAdd(c.d_done, (unsigned long long)n);
    if ((old >> 8) != ((old + n) >> 8))
        __hip_atomic_store(c.h_done, old + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// hipFuncSetAttribute is per device: "done once" flags are kept per device
struct PerDeviceOnce {
    static constexpr int MAX_DEV = 64;
    bool done[MAX_DEV] = {};
    bool &flag()
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        return done[(dev >= 0 && dev < MAX_DEV) ? dev : 0];
    }
};

// XCD-aware remap of a 1-D block index: consecutive *logical* blocks land on
// the same XCD (dispatcher places physical block b on XCD b % 8), so blocks
// that share operand panels share an L2.  Bijective for any grid size
// (cdna_hip_programming.md, "XCD swizzle must be bijective").
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg)
{
    const unsigned nx = 8;
    unsigned q = nwg / nx, r = nwg % nx;
    unsigned xcd = bid % nx, pos = bid / nx;
    unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + pos;
}

}  // namespace lk

// Host object behind the opaque lk_task_ctl handle (misc.hip).
struct lk_task_ctl {
    int *h_words = nullptr;  // pinned + mapped: [0] cancel, [2..3] done (64-bit)
    int *dh_words = nullptr;  // device address of h_words
    int *d_words = nullptr;  // HBM: [0] cancel seen, [2..3] done (64-bit)
    int64_t rows_total = 0;   // what progress is reported in (rows)
    int64_t units_total = 0;  // what the kernels count (rows, or tasks of several per row)
    lk::TaskCtlDev dev() const
    {
        lk::TaskCtlDev d;
        d.h_cancel = dh_words;
        d.h_done = reinterpret_cast<unsigned long long *>(dh_words + 2);
        d.d_cancel = d_words;
        d.d_done = reinterpret_cast<unsigned long long *>(d_words + 2);
        return d;
    }
};

namespace lk {
// full descending sort of score rows (topn_sort.hip): the n = None / n > 4096 path of top-N
size_t topn_sort_workspace_bytes(int64_t n_rows, int64_t row_len);
int topn_sort(const float *scores, int64_t ld_s, int64_t n_rows, int64_t row_len, int64_t n,
              void *ws, int32_t *out_idx, float *out_score, int64_t out_ld, hipStream_t st);
// start of a controlled call: zero the device counters, record the units (asynchronous)
int ctl_begin(lk_task_ctl *ctl, int64_t rows_total, int64_t units_total, hipStream_t st);
// after the stream is synchronised: LK_E_CANCELLED if the kernels saw the cancel; posts the
// exact final count to the host word
int ctl_finish(lk_task_ctl *ctl, hipStream_t st);
}  // namespace lk
