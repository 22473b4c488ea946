// Probe of v_permlane32_swap / v_permlane16_swap lane semantics on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *o)
{
    unsigned lane = threadIdx.x;
    unsigned x = lane, y = 100 + lane;
    auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    o[lane] = r[0]; o[64 + lane] = r[1];
    auto s = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    o[128 + lane] = s[0]; o[192 + lane] = s[1];
}
int main()
{
    unsigned *d; (void)hipMalloc(&d, 1024);
    k<<<1, 64>>>(d);
    unsigned h[256]; (void)hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    const char *n[4] = { "swap32 r0", "swap32 r1", "swap16 r0", "swap16 r1" };
    for (int a = 0; a < 4; a++) { printf("%s:", n[a]); for (int l = 0; l < 64; l += 4) printf(" %u", h[64 * a + l]); printf("\n"); }
    return 0;
}
