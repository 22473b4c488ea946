// Microbenchmark: fp32 FMA rate of v_fma_f32 against v_pk_fma_f32 on gfx950, 8 waves per SIMD, 8 independent chains per lane.
//   hipcc -O3 --offload-arch=gfx950 -o build/mb/pk tools/mb_pk_fma.hip && ./build/mb/pk      (profiles/r03_mb_pk_fma.log)
// Round 3 measured 96-103 TFLOP/s with v_fma_f32 and 110-119 with v_pk_fma_f32: the packed form buys 1.16x, not 2x, so
// writing the compositing passes on pairs of entries (tried: sg_render.hip history, 154 vs 147 us at cfg3) cannot pay.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, float a, float b, int iters)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    f2 y0 = {x0, x1}, y1 = {x2, x3}, y2 = {x4, x5}, y3 = {x6, x7};
    f2 a2 = {a, a}, b2 = {b, b};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                             "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                y0 = __builtin_elementwise_fma(y0, a2, b2); y1 = __builtin_elementwise_fma(y1, a2, b2);
                y2 = __builtin_elementwise_fma(y2, a2, b2); y3 = __builtin_elementwise_fma(y3, a2, b2);
            }
        }
    }
    if (MODE == 0) out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    else out[blockIdx.x * 256 + threadIdx.x] = y0.x + y0.y + y1.x + y1.y + y2.x + y2.y + y3.x + y3.y;
}
int main()
{
    float *out; hipMalloc(&out, 256 * 2048 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, grid = 256 * 8;   // 8 workgroups per CU = 8 waves per SIMD
    for (int mode = 0; mode < 2; mode++)
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters);
            else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fma = (double)grid * 256 * iters * 64;      // FMAs executed (both modes: 64 per thread per iteration)
            printf("mode %s: %.3f ms, %.1f TFLOP/s (fp32 FMA x2)\n", mode ? "v_pk_fma_f32" : "v_fma_f32", ms, 2 * fma / ms * 1e-9);
        }
    return 0;
}
