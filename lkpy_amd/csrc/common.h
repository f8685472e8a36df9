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

// broadcast lane `src` (compile-time or wave-uniform) of v
__device__ __forceinline__ float bcast(float v, int src)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}

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
