// Probe: where do the totals of sg_reduce9 land?  (hipcc --offload-arch=gfx950 tools/reduce9_probe.hip)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define SG_DPP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, false))
__device__ __forceinline__ float sg_fold32(float x, float y)
{
    // inline asm: hipcc (ROCm 7.2) folds "r[0] + r[1]" of __builtin_amdgcn_permlane32_swap into r[0] + r[0]
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    return x + y;
}
__device__ __forceinline__ float sg_fold16(float x, float y)
{
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    return x + y;
}
__global__ void k(float *out, float *raw)
{
    int lane = threadIdx.x;
    float v[9];
    for (int i = 0; i < 9; i++) v[i] = (float)((i + 1) * 1000) + (float)((lane * 7 + i * 3) % 13);
    float s01 = sg_fold32(v[0], v[1]), s23 = sg_fold32(v[2], v[3]);
    float s45 = sg_fold32(v[4], v[5]), s67 = sg_fold32(v[6], v[7]);
    raw[lane] = s01; raw[64 + lane] = s23;
    float t0 = sg_fold16(s01, s23);
    float t1 = sg_fold16(s45, s67);
    raw[128 + lane] = t0;
    float u = t0 + SG_DPP(t0, 0x128);
    float w = t1 + SG_DPP(t1, 0x128);
    float z = (lane & 8) ? w : u;
    z += SG_DPP(z, 0xB1);
    z += SG_DPP(z, 0x4E);
    z += SG_DPP(z, 0x141);
    out[lane] = z;
}
int main()
{
    float *d, *r; hipMalloc(&d, 256); hipMalloc(&r, 192 * 4);
    k<<<1, 64>>>(d, r);
    float h[64], hr[192]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost); hipMemcpy(hr, r, 768, hipMemcpyDeviceToHost);
    for (int g = 0; g < 8; g++) {
        int idx = (int)(h[g * 8] / 64 / 1000) - 1; double ex = 0;
        for (int l = 0; l < 64; l++) ex += (idx + 1) * 1000 + ((l * 7 + idx * 3) % 13);
        printf("group %d: total %.0f -> value index %d expected %.0f %s\n", g, h[g * 8], idx, ex, ex == h[g * 8] ? "OK" : "MISMATCH");
    }
    printf("s01: lane0 %.0f lane32 %.0f | s23: lane0 %.0f lane32 %.0f\n", hr[0], hr[32], hr[64], hr[96]);
    printf("t0 rows: %.0f %.0f %.0f %.0f\n", hr[128], hr[128 + 16], hr[128 + 32], hr[128 + 48]);
    return 0;
}
