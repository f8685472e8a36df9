// Probe: how fast does a CU's LDS serve 64-lane atomics?  Eight waves per CU (two workgroups of 4,
// 16 KiB of cells per wave: the accumulating item-kNN kernel's residency), every wave issues N
// back-to-back LDS operations on its own 4096 cells; reports CU cycles per wave instruction for
//   ds_add_f32 / ds_add_u32 / ds_write_b32 / ds_read_b32  x  (distinct consecutive cells, random
//   cells, 8 lanes per cell).
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomic_rate lds_atomic_rate.hip && ./lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int CELLS = 4096, WAVES = 4, N = 4096;

template <int OP>
__global__ __launch_bounds__(256) void k(const int *__restrict__ addr, unsigned long long *out, float *sink)
{
    __shared__ unsigned cell[WAVES][CELLS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned *c = cell[wave];
    for (int i = lane; i < CELLS; i += 64) c[i] = 0u;
    __syncthreads();
    int a[8];
    for (int u = 0; u < 8; ++u) a[u] = addr[(blockIdx.x * 8 + u) * 64 + lane];
    float acc = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N / 8; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int x = (a[u] + i * 67) & (CELLS - 1);
            if (OP == 0) (void)__hip_atomic_fetch_add(reinterpret_cast<float *>(&c[x]), 1.5f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (OP == 1) (void)__hip_atomic_fetch_add(&c[x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (OP == 2) c[x] = (unsigned)i;
            if (OP == 3) acc += __builtin_bit_cast(float, c[x]);
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 123.456f) sink[0] = acc + __builtin_bit_cast(float, c[lane]);
}

int main()
{
    const int blocks = 512;  // two per CU
    std::vector<int> h(blocks * 8 * 64);
    int *d;
    unsigned long long *out;
    float *sink;
    hipMalloc(&d, h.size() * 4);
    hipMalloc(&out, blocks * 8);
    hipMalloc(&sink, 4);
    const char *pat[3] = {"consecutive", "random", "8 lanes per cell"};
    const char *ops[4] = {"ds_add_f32", "ds_add_u32", "ds_write_b32", "ds_read_b32"};
    for (int p = 0; p < 3; ++p) {
        srand(7);
        for (size_t i = 0; i < h.size(); ++i) {
            const int lane = (int)(i & 63);
            h[i] = p == 0 ? lane + 64 * (int)((i >> 6) & 7) : p == 1 ? rand() & (CELLS - 1) : ((rand() & (CELLS - 1)) & ~7) ;
            if (p == 2) h[i] = (((int)(i >> 6) * 997) & (CELLS - 1) & ~7) + (lane >> 3);
        }
        hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int op = 0; op < 4; ++op) {
            for (int rep = 0; rep < 2; ++rep) {
                if (op == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, out, sink);
                if (op == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, out, sink);
                if (op == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, out, sink);
                if (op == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, out, sink);
            }
            std::vector<unsigned long long> o(blocks);
            hipMemcpy(o.data(), out, blocks * 8, hipMemcpyDeviceToHost);
            double s = 0;
            for (auto v : o) s += (double)v;
            // a CU holds 8 waves (two workgroups): N instructions each in `cycles` -> cycles per wave
            // instruction of the CU = cycles / (8 N)
            printf("%-13s %-17s  %7.0f cycles per workgroup for %d ops per wave -> %.1f CU cycles per wave instruction (%.1f lanes per clock)\n",
                   ops[op], pat[p], s / blocks, N, s / blocks / (8.0 * N), 64.0 / (s / blocks / (8.0 * N)));
        }
    }
    return 0;
}
