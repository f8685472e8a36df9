// Microbenchmark: LDS float-atomic and slice-load rates at one wave per SIMD (the
// occupancy of iknn_build_kernel).  hipcc --offload-arch=gfx950 -O3 -o /tmp/ub lds_atomic.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// each wave: ITER x { ds_add_f32 with `active` lanes at pseudo-random cells }
template <int MODE>
__global__ __launch_bounds__(256) void k_atomic(const int *__restrict__ cells, int iters, int active, float *out)
{
    extern __shared__ float acc[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *a = acc + wave * (8192 + 64);
    for (int c = lane; c < 8192 + 64; c += 64) a[c] = 0.f;
    int cell = cells[(blockIdx.x * 256 + threadIdx.x) & 0xffff];
    for (int i = 0; i < iters; ++i) {
        cell = (cell * 1103515245 + 12345) & 0x7fffffff;
        const int c = (cell >> 8) & 8191;
        if (MODE == 0) {          // exec-masked atomic
            if (lane < active) __hip_atomic_fetch_add(&a[c], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        } else if (MODE == 1) {   // all lanes, idle ones to a private dummy cell
            const int cc = lane < active ? c : 8192 + lane;
            __hip_atomic_fetch_add(&a[cc], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        } else if (MODE == 2) {   // plain read-modify-write (not atomic; rate reference)
            if (lane < active) a[c] += 1.0f;
        } else {                  // ALU only
            if (lane < active) a[0] = (float)c;
        }
    }
    float s = 0;
    for (int c = lane; c < 8192; c += 64) s += a[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// each wave: ITER x { 8-byte/lane load of a random `len`-entry slice, RING in flight }
template <int RING>
__global__ __launch_bounds__(256) void k_load(const int2 *__restrict__ pack, const int *__restrict__ starts, int n_starts,
                                               int iters, int len, float *out)
{
    const int lane = threadIdx.x & 63;
    const int gw = (blockIdx.x * 256 + threadIdx.x) >> 6;
    int2 ring[RING];
    float s = 0;
    int pos = (gw * 977) % n_starts;
    const int li = min(lane, len - 1);
#pragma unroll
    for (int q = 0; q < RING; ++q) {
        const int st = __builtin_amdgcn_readfirstlane(starts[(pos + q) % n_starts]);
        ring[q] = pack[st + li];
    }
    for (int i = 0; i < iters; i += RING) {
#pragma unroll
        for (int q = 0; q < RING; ++q) {
            const int2 e = ring[q];
            const int st = __builtin_amdgcn_readfirstlane(starts[(pos + i + q + RING) % n_starts]);
            ring[q] = pack[st + li];
            s += __builtin_bit_cast(float, e.y) + (float)e.x;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    const int blocks = 256, iters = 20000;
    std::vector<int> h(65536);
    for (auto &x : h) x = rand();
    int *cells; float *out;
    CK(hipMalloc(&cells, 65536 * 4)); CK(hipMalloc(&out, blocks * 256 * 4));
    CK(hipMemcpy(cells, h.data(), 65536 * 4, hipMemcpyHostToDevice));
    const size_t lds = 4 * (8192 + 64) * 4;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](auto kern, const char *name, int active) {
        CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, cells, 100, active, out);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, cells, iters, active, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s active=%2d  %.3f ms  -> %.1f cycles/iter/wave (2.4 GHz)\n", name, active, ms, ms * 1e-3 * 2.4e9 / iters);
    };
    for (int active : {8, 19, 32, 64}) {
        run(k_atomic<0>, "atomic exec-masked", active);
        run(k_atomic<1>, "atomic dummy-cell", active);
        run(k_atomic<2>, "plain rmw", active);
        run(k_atomic<3>, "alu only", active);
    }
    // slice loads: 200 MB pack, random starts
    const int64_t nnz = 25000000; const int n_starts = 1 << 22;
    int2 *pack; int *starts;
    CK(hipMalloc(&pack, (nnz + 64) * 8)); CK(hipMemset(pack, 0, (nnz + 64) * 8));
    std::vector<int> hs(n_starts);
    for (auto &x : hs) x = (int)(((int64_t)rand() * 32768 + rand()) % (nnz - 64));
    CK(hipMalloc(&starts, n_starts * 4)); CK(hipMemcpy(starts, hs.data(), n_starts * 4, hipMemcpyHostToDevice));
    auto runl = [&](auto kern, const char *name, int len, int blocks_) {
        hipLaunchKernelGGL(kern, dim3(blocks_), dim3(256), 0, 0, pack, starts, n_starts, 64, len, out);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks_), dim3(256), 0, 0, pack, starts, n_starts, 4096, len, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double loads = (double)blocks_ * 4 * 4096;
        printf("%-16s len=%2d blocks=%4d  %.3f ms -> %.1f cycles/load/wave, %.2f Gloads/s, %.1f GB/s useful\n", name, len, blocks_, ms,
               ms * 1e-3 * 2.4e9 / 4096, loads / ms * 1e-6, loads * len * 8 / ms * 1e-6);
    };
    CK(hipFree(out)); CK(hipMalloc(&out, 1024 * 256 * 4));
    for (int len : {19, 64}) {
        runl(k_load<1>, "ring1", len, 256);
        runl(k_load<4>, "ring4", len, 256);
        runl(k_load<8>, "ring8", len, 256);
        runl(k_load<16>, "ring16", len, 256);
        runl(k_load<8>, "ring8", len, 1024);
    }
    return 0;
}
