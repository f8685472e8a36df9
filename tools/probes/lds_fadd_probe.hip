// Probe: are same-address LDS float adds (ds_add_f32, no return) of ONE instruction applied in
// ascending lane order, with IEEE round-to-nearest results equal to the VALU's sequential sums?
//   hipcc --offload-arch=gfx950 -O3 -o lds_fadd_probe lds_fadd_probe.hip && ./lds_fadd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#pragma clang fp contract(off)

__global__ void probe(const float *__restrict__ vals, const int *__restrict__ addr, int rounds,
                      float *__restrict__ out)
{
    __shared__ float cell[64];
    const int lane = threadIdx.x;
    cell[lane] = 0.f;
    __syncthreads();
    for (int r = 0; r < rounds; ++r) {
        const float v = vals[r * 64 + lane];
        const int a = addr[r * 64 + lane];
        if (a >= 0) __hip_atomic_fetch_add(&cell[a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    out[lane] = cell[lane];
}

int main()
{
    const int rounds = 4096, trials = 16;
    int bad_total = 0;
    for (int trial = 0; trial < trials; ++trial) {
        std::vector<float> v(rounds * 64);
        std::vector<int> a(rounds * 64);
        srand(1234 + trial);
        for (int r = 0; r < rounds; ++r) {
            const int pat = rand() % 8;
            for (int l = 0; l < 64; ++l) {
                int ad;
                switch (pat) {
                    case 0: ad = 0; break;
                    case 1: ad = l & 1; break;
                    case 2: ad = l % 5; break;
                    case 3: ad = (l * 7) % 13; break;
                    case 4: ad = l >> 4; break;
                    case 5: ad = rand() % 64; break;
                    case 6: ad = l & 31; break;
                    default: ad = rand() % 3; break;
                }
                if (rand() % 7 == 0) ad = -1;
                a[r * 64 + l] = ad;
                float x;
                const int kind = trial % 4;
                if (kind == 0) x = (float)rand() / RAND_MAX;                       // similarities
                else if (kind == 1) x = ((float)rand() / RAND_MAX - 0.5f) * 5.f;   // w * centred rating
                else if (kind == 2) x = ldexpf((float)rand() / RAND_MAX, rand() % 40 - 20) * ((rand() & 1) ? 1 : -1);
                else x = ldexpf((float)rand() / RAND_MAX, -140 + rand() % 20) * ((rand() & 1) ? 1 : -1);  // denormal range
                v[r * 64 + l] = x;
            }
        }
        std::vector<float> want(64, 0.f);
        for (int r = 0; r < rounds; ++r)
            for (int l = 0; l < 64; ++l)
                if (a[r * 64 + l] >= 0) {
                    volatile float s = want[a[r * 64 + l]] + v[r * 64 + l];
                    want[a[r * 64 + l]] = s;
                }
        float *dv, *dout;
        int *da;
        hipMalloc(&dv, v.size() * 4);
        hipMalloc(&da, a.size() * 4);
        hipMalloc(&dout, 64 * 4);
        hipMemcpy(dv, v.data(), v.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dv, da, rounds, dout);
        std::vector<float> got(64);
        hipMemcpy(got.data(), dout, 64 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            if (memcmp(&got[l], &want[l], 4) != 0) {
                if (bad < 3) printf("  trial %d cell %d: got %.9g want %.9g\n", trial, l, got[l], want[l]);
                ++bad;
            }
        printf("trial %d (kind %d): %d of 64 cells differ\n", trial, trial % 4, bad);
        bad_total += bad;
        hipFree(dv); hipFree(da); hipFree(dout);
    }
    printf("TOTAL differing cells: %d\n", bad_total);
    return 0;
}
