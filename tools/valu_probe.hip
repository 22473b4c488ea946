// VALU issue-rate probe for gfx950: cycles per wave64 instruction for several instruction kinds, as a function of
// resident waves per SIMD and of the number of independent dependency chains per wave.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_probe.hip -o tools/valu_probe && ./tools/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(X) X X X X X X X X
template <int MODE>
__global__ void __launch_bounds__(256) probe(float *out, int iters)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = { a0, a1 }, p1 = { a2, a3 }, p2 = { a4, a5 }, p3 = { a6, a7 };
    const float m = 1.0001f, c = 0.5f;
    unsigned r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    const unsigned long long k0 = threadIdx.x * 0x100000001ull, k1 = k0 + 77, k2 = k0 * 3, k3 = k0 + (7ull << 33), key = 0x0000004000000040ull;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {        // 8 independent fma chains, 64 instr
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
        } else if (MODE == 1) { // 1 dependent fma chain, 64 instr
            REP8(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                              "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2"
                              : "+v"(a0) : "v"(m), "v"(c));)
        } else if (MODE == 2) { // 2 dependent chains interleaved
            REP8(asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                              "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3"
                              : "+v"(a0), "+v"(a1) : "v"(m), "v"(c));)
        } else if (MODE == 3) { // 4 packed fma chains
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                              "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p0), "v"(p1));)
        } else if (MODE == 4) { // exp, 8 chains
            REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                              "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 5) { // dpp add, 8 chains
            REP8(asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 6) { // v_mul_f32 e32, 8 chains
            REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                              "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (MODE == 7) { // cmp + cndmask pairs
            REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %2, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %4, %4, %5, vcc\n"
                              "v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %0, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %6, vcc"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");)
        } else if (MODE == 8) { // permlane32 swap, 4 pairs
            REP8(asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                              "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 9) { // 64-bit compare + add-with-carry pairs (the rank sort of sg_sort.h counts keys this way)
            REP8(asm volatile("v_cmp_lt_u64 vcc, %4, %8\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n v_cmp_lt_u64 vcc, %5, %8\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n"
                              "v_cmp_lt_u64 vcc, %6, %8\n v_addc_co_u32 %2, vcc, 0, %2, vcc\n v_cmp_lt_u64 vcc, %7, %8\n v_addc_co_u32 %3, vcc, 0, %3, vcc"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(k0), "v"(k1), "v"(k2), "v"(k3), "v"(key) : "vcc");)
        } else if (MODE == 10) { // the same with 32-bit compares
            REP8(asm volatile("v_cmp_lt_u32 vcc, %4, %8\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n v_cmp_lt_u32 vcc, %5, %8\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n"
                              "v_cmp_lt_u32 vcc, %6, %8\n v_addc_co_u32 %2, vcc, 0, %2, vcc\n v_cmp_lt_u32 vcc, %7, %8\n v_addc_co_u32 %3, vcc, 0, %3, vcc"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"((unsigned)k0), "v"((unsigned)k1), "v"((unsigned)k2), "v"((unsigned)k3), "v"((unsigned)key) : "vcc");)
        }
    }
    a0 += (float)(r0 + r1 + r2 + r3);
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE> static void run(const char *name, float *out)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    printf("%-28s", name);
    for (int wps = 1; wps <= 8; wps *= 2) {       // waves per SIMD: one 256-thread workgroup = 1 wave on each SIMD of a CU
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(probe<MODE>, dim3(256 * wps), dim3(256), 0, 0, out, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        printf("  %dw: %5.2f", wps, ms * 1e6 / ((double)iters * 64 * wps) * 2.4);
    }
    printf("   (cycles @2.4 GHz per wave-instruction per SIMD)\n");
}
int main()
{
    float *out; (void)hipMalloc(&out, 256 * 2048 * 4 * 2);
    run<0>("v_fma_f32 8 chains", out); run<2>("v_fma_f32 2 chains", out); run<1>("v_fma_f32 1 chain", out);
    run<6>("v_mul_f32 8 chains", out); run<3>("v_pk_fma_f32 4 chains", out); run<4>("v_exp_f32 8 chains", out);
    run<5>("v_add_f32_dpp 8 chains", out); run<7>("v_cmp+v_cndmask", out); run<8>("v_permlane32_swap", out);
    run<9>("v_cmp_lt_u64 + v_addc", out); run<10>("v_cmp_lt_u32 + v_addc", out);
    return 0;
}
