// Does fp32 MFMA 4x4x1 (16 blocks) overlap with plain fp32 VALU on gfx950?  Cycles per group per SIMD for: MFMA only, VALU only (3 v_fmac), both.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_coissue_probe.hip -o tools/mfma_coissue_probe && ./tools/mfma_coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
#define REP8(X) X X X X X X X X
template <int MODE, int NV>
__global__ void __launch_bounds__(256) probe(float *out, unsigned long long *cyc, int iters)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, m = 1.0001f;
    float b0 = a0 * 3, b1 = a0 * 5, b2 = a0 * 7, b3 = a0 * 9;   // MFMA operands: never written in the loop (no VALU -> MFMA hazard)
    asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
    f4 d0 = { 0, 0, 0, 0 }, d1 = d0, d2 = d0, d3 = d0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        REP8(
            if (MODE & 1) {
                d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(b0, m, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(b1, m, d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_f32_4x4x1f32(b2, m, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_f32_4x4x1f32(b3, m, d3, 0, 0, 0);
            }
            if (MODE & 2) {
                for (int r = 0; r < NV; r++)
                    asm volatile("v_fmac_f32 %0, %6, %1\n v_fmac_f32 %1, %6, %2\n v_fmac_f32 %2, %6, %3\n v_fmac_f32 %3, %6, %4\n v_fmac_f32 %4, %6, %5\n v_fmac_f32 %5, %6, %0"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(m));
            })
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + d0.x + d1.y + d2.z + d3.w + d0.y + d1.x;
    if ((threadIdx.x & 63) == 0) { unsigned long long *q = cyc + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4; q[0] = t0; q[1] = t1; q[2] = r0; q[3] = r1; }
}
template <int MODE, int NV> static void run(const char *name, float *out, unsigned long long *cyc)
{
    const int iters = 2000;
    printf("%-40s", name);
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int nb = 256 * wps;
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL((probe<MODE, NV>), dim3(nb), dim3(256), 0, 0, out, cyc, iters); (void)hipDeviceSynchronize(); }
        static unsigned long long h[256 * 4 * 4 * 4];
        (void)hipMemcpy(h, cyc, sizeof(unsigned long long) * nb * 16, hipMemcpyDeviceToHost);
        double clk = 0; unsigned long long rmin = ~0ull, rmax = 0;
        for (int i = 0; i < nb * 4; i++) {
            clk += (double)(h[4 * i + 1] - h[4 * i]) / (double)(h[4 * i + 3] - h[4 * i + 2]) * 0.1;
            if (h[4 * i + 2] < rmin) rmin = h[4 * i + 2];
            if (h[4 * i + 3] > rmax) rmax = h[4 * i + 3];
        }
        clk /= nb * 4;
        printf("  %dw: %6.1f cyc/group @%.2f GHz", wps, (double)(rmax - rmin) * 10.0 * clk / ((double)iters * 8 * wps), clk);
    }
    printf("\n");
}
int main()
{
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 256 * 1024 * 4 * 2); (void)hipMalloc(&cyc, 256 * 4 * 4 * 8 * 4);
    printf("group = 4 x v_mfma_f32_4x4x1_16b_f32 and / or NV x 6 v_fmac_f32 (VGPR sources); cycles per group per SIMD\n");
    run<1, 0>("4 MFMA", out, cyc); run<2, 2>("12 v_fmac", out, cyc); run<3, 2>("4 MFMA + 12 v_fmac", out, cyc);
    run<2, 4>("24 v_fmac", out, cyc); run<3, 4>("4 MFMA + 24 v_fmac", out, cyc);
    return 0;
}
