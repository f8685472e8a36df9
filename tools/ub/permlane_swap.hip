// Semantics check of v_permlane32_swap / v_permlane16_swap (gfx950) as als_chol.hip uses them:
// prints, for the 4 x 4 (register x row-group) transposition, which (register, group) each
// output came from.  hipcc --offload-arch=gfx950 -O2 tools/ub/permlane_swap.hip -o permlane_swap
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k(unsigned *out)
{
    const unsigned lane = threadIdx.x;
    unsigned x[4];
    for (int r = 0; r < 4; ++r) x[r] = (r << 8) | lane;  // register r, lane
    // the statements als_chol.hip uses (chained __builtin_amdgcn_permlane*_swap calls are
    // miscompiled by hipcc 7.2: all four results come out of one register)
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x[0]), "+v"(x[2]));
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x[1]), "+v"(x[3]));
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x[0]), "+v"(x[1]));
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x[2]), "+v"(x[3]));
    for (int r = 0; r < 4; ++r) out[r * 64 + lane] = x[r];
}

int main()
{
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int r = 0; r < 4; ++r)
        for (int l = 0; l < 64; ++l) {
            // expected: y[r] at lane (g, c) = x[g] at lane (r, c)
            const unsigned want = ((l >> 4) << 8) | (r * 16 + (l & 15));
            if (h[r * 64 + l] != want) ++bad;
        }
    printf("permlane 4x4 transpose: %s (%d mismatches)\n", bad ? "DIFFERENT" : "as assumed", bad);
    if (bad)
        for (int r = 0; r < 4; ++r) {
            for (int g = 0; g < 4; ++g) printf(" y[%d]@g%d = x[%u]@lane%u", r, g, h[r * 64 + g * 16] >> 8, h[r * 64 + g * 16] & 255);
            printf("\n");
        }
    return bad != 0;
}
