// VALU issue cost on gfx950 by OPERAND KIND: cycles (shader clock, s_memtime) per wave64 instruction per SIMD for fp32 fma / mac streams
// whose sources are (a) VGPRs only, (b) one SGPR, (c) many distinct VGPRs (a 55-register ring, as in sg_loss.hip), (d) an inline constant.
//   hipcc -O3 --offload-arch=gfx950 tools/fmac_probe.hip -o tools/fmac_probe && ./tools/fmac_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(X) X X X X
#define REP8(X) X X X X X X X X
template <int MODE>
__global__ void __launch_bounds__(256) probe(float *out, unsigned long long *cyc, int iters, float sw)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float r[48];
#pragma unroll
    for (int i = 0; i < 48; i++) r[i] = a0 * (i + 1);
    const float m = 1.0001f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {        // v_fmac v,v,v : 8 chains, 64 instr
            REP8(asm volatile("v_fmac_f32 %0, %8, %1\n v_fmac_f32 %1, %8, %2\n v_fmac_f32 %2, %8, %3\n v_fmac_f32 %3, %8, %4\n"
                              "v_fmac_f32 %4, %8, %5\n v_fmac_f32 %5, %8, %6\n v_fmac_f32 %6, %8, %7\n v_fmac_f32 %7, %8, %0"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (MODE == 1) { // v_fmac v, s, v : SGPR multiplier
            REP8(asm volatile("v_fmac_f32 %0, %8, %1\n v_fmac_f32 %1, %8, %2\n v_fmac_f32 %2, %8, %3\n v_fmac_f32 %3, %8, %4\n"
                              "v_fmac_f32 %4, %8, %5\n v_fmac_f32 %5, %8, %6\n v_fmac_f32 %6, %8, %7\n v_fmac_f32 %7, %8, %0"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sw));)
        } else if (MODE == 2) { // v_mul v, s, v
            REP8(asm volatile("v_mul_f32 %0, %8, %1\n v_mul_f32 %1, %8, %2\n v_mul_f32 %2, %8, %3\n v_mul_f32 %3, %8, %4\n"
                              "v_mul_f32 %4, %8, %5\n v_mul_f32 %5, %8, %6\n v_mul_f32 %6, %8, %7\n v_mul_f32 %7, %8, %0"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sw));)
        } else if (MODE == 3) { // v_fmac acc, s, ring[k]: 5 chains over 40 distinct ring registers, as the vertical window of sg_loss.hip
            REP4(asm volatile("v_fmac_f32 %0, %5, %6\n v_fmac_f32 %1, %5, %7\n v_fmac_f32 %2, %5, %8\n v_fmac_f32 %3, %5, %9\n v_fmac_f32 %4, %5, %10\n"
                              "v_fmac_f32 %0, %5, %11\n v_fmac_f32 %1, %5, %12\n v_fmac_f32 %2, %5, %13\n v_fmac_f32 %3, %5, %14\n v_fmac_f32 %4, %5, %15\n"
                              "v_fmac_f32 %0, %5, %16\n v_fmac_f32 %1, %5, %17\n v_fmac_f32 %2, %5, %18\n v_fmac_f32 %3, %5, %19\n v_fmac_f32 %4, %5, %20\n v_fmac_f32 %0, %5, %21"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4) : "s"(sw), "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]),
                                "v"(r[6]), "v"(r[7]), "v"(r[8]), "v"(r[9]), "v"(r[10]), "v"(r[11]), "v"(r[12]), "v"(r[13]), "v"(r[14]), "v"(r[15]));)
        } else if (MODE == 4) { // the same with the multiplier in a VGPR
            REP4(asm volatile("v_fmac_f32 %0, %5, %6\n v_fmac_f32 %1, %5, %7\n v_fmac_f32 %2, %5, %8\n v_fmac_f32 %3, %5, %9\n v_fmac_f32 %4, %5, %10\n"
                              "v_fmac_f32 %0, %5, %11\n v_fmac_f32 %1, %5, %12\n v_fmac_f32 %2, %5, %13\n v_fmac_f32 %3, %5, %14\n v_fmac_f32 %4, %5, %15\n"
                              "v_fmac_f32 %0, %5, %16\n v_fmac_f32 %1, %5, %17\n v_fmac_f32 %2, %5, %18\n v_fmac_f32 %3, %5, %19\n v_fmac_f32 %4, %5, %20\n v_fmac_f32 %0, %5, %21"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4) : "v"(m), "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]),
                                "v"(r[6]), "v"(r[7]), "v"(r[8]), "v"(r[9]), "v"(r[10]), "v"(r[11]), "v"(r[12]), "v"(r[13]), "v"(r[14]), "v"(r[15]));)
        } else if (MODE == 5) { // v_fma v, v, v, v (VOP3, three VGPR sources + separate destination), 8 chains
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %1\n v_fma_f32 %1, %1, %8, %2\n v_fma_f32 %2, %2, %8, %3\n v_fma_f32 %3, %3, %8, %4\n"
                              "v_fma_f32 %4, %4, %8, %5\n v_fma_f32 %5, %5, %8, %6\n v_fma_f32 %6, %6, %8, %7\n v_fma_f32 %7, %7, %8, %0"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (MODE == 6) { // v_pk_fma_f32, 4 chains of pairs, VGPR sources
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = { a0, a1 }, p1 = { a2, a3 }, p2 = { a4, a5 }, p3 = { a6, a7 }, mm = { m, m };
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %1\n v_pk_fma_f32 %1, %1, %4, %2\n v_pk_fma_f32 %2, %2, %4, %3\n v_pk_fma_f32 %3, %3, %4, %0\n"
                              "v_pk_fma_f32 %0, %0, %4, %1\n v_pk_fma_f32 %1, %1, %4, %2\n v_pk_fma_f32 %2, %2, %4, %3\n v_pk_fma_f32 %3, %3, %4, %0"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(mm));)
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
        } else if (MODE == 8) { // inline constant source
            REP8(asm volatile("v_mul_f32 %0, 2.0, %1\n v_mul_f32 %1, 2.0, %2\n v_mul_f32 %2, 2.0, %3\n v_mul_f32 %3, 2.0, %4\n"
                              "v_mul_f32 %4, 2.0, %5\n v_mul_f32 %5, 2.0, %6\n v_mul_f32 %6, 2.0, %7\n v_mul_f32 %7, 2.0, %0"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 9) { // 32-bit literal source
            REP8(asm volatile("v_add_f32 %0, 0x38d1b717, %1\n v_add_f32 %1, 0x38d1b717, %2\n v_add_f32 %2, 0x38d1b717, %3\n v_add_f32 %3, 0x38d1b717, %4\n"
                              "v_add_f32 %4, 0x38d1b717, %5\n v_add_f32 %5, 0x38d1b717, %6\n v_add_f32 %6, 0x38d1b717, %7\n v_add_f32 %7, 0x38d1b717, %0"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 10) { // v_cndmask with vcc (set once)
            asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a0), "v"(a1) : "vcc");
            REP8(asm volatile("v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n v_cndmask_b32 %2, %3, %4, vcc\n v_cndmask_b32 %3, %4, %5, vcc\n"
                              "v_cndmask_b32 %4, %5, %6, vcc\n v_cndmask_b32 %5, %6, %7, vcc\n v_cndmask_b32 %6, %7, %0, vcc\n v_cndmask_b32 %7, %0, %1, vcc"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");)
        } else if (MODE == 11) { // v_cmp writing vcc
            REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %4\n"
                              "v_cmp_lt_f32 vcc, %4, %5\n v_cmp_lt_f32 vcc, %5, %6\n v_cmp_lt_f32 vcc, %6, %7\n v_cmp_lt_f32 vcc, %7, %0"
                              :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "vcc");)
        } else if (MODE == 12) { // v_max_f32 / v_min_f32 (VOP2, VGPRs)
            REP8(asm volatile("v_max_f32 %0, %0, %1\n v_min_f32 %1, %1, %2\n v_max_f32 %2, %2, %3\n v_min_f32 %3, %3, %4\n"
                              "v_max_f32 %4, %4, %5\n v_min_f32 %5, %5, %6\n v_max_f32 %6, %6, %7\n v_min_f32 %7, %7, %0"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 13) { // v_mov_b32
            REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                              "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 14) { // v_add_u32 / v_lshlrev (integer)
            REP8(asm volatile("v_add_u32 %0, %0, %1\n v_lshlrev_b32 %1, 1, %2\n v_add_u32 %2, %2, %3\n v_and_b32 %3, %3, %4\n"
                              "v_add_u32 %4, %4, %5\n v_lshlrev_b32 %5, 1, %6\n v_add_u32 %6, %6, %7\n v_and_b32 %7, %7, %0"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 7) { // v_add_f32 v, v, v
            REP8(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %4\n"
                              "v_add_f32 %4, %4, %5\n v_add_f32 %5, %5, %6\n v_add_f32 %6, %6, %7\n v_add_f32 %7, %7, %0"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float acc = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
#pragma unroll
    for (int i = 0; i < 48; i++) acc += r[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) { unsigned long long *q = cyc + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4; q[0] = t0; q[1] = t1; q[2] = r0; q[3] = r1; }
}
template <int MODE> static void run(const char *name, float *out, unsigned long long *cyc)
{
    const int iters = 2000, per = 64;
    printf("%-44s", name);
    for (int wps = 1; wps <= 4; wps *= 2) {       // waves per SIMD: one 256-thread workgroup = 1 wave on each SIMD of a CU
        const int nb = 256 * wps;
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(probe<MODE>, dim3(nb), dim3(256), 0, 0, out, cyc, iters, 1.0001f);
            (void)hipDeviceSynchronize();
        }
        static unsigned long long h[256 * 8 * 4 * 4];
        (void)hipMemcpy(h, cyc, sizeof(unsigned long long) * nb * 16, hipMemcpyDeviceToHost);
        double clk = 0; unsigned long long rmin = ~0ull, rmax = 0;
        for (int i = 0; i < nb * 4; i++) {
            clk += (double)(h[4 * i + 1] - h[4 * i]) / (double)(h[4 * i + 3] - h[4 * i + 2]) * 0.1;    // GHz (100 MHz real-time clock)
            if (h[4 * i + 2] < rmin) rmin = h[4 * i + 2];
            if (h[4 * i + 3] > rmax) rmax = h[4 * i + 3];
        }
        clk /= nb * 4;
        const double wall_ns = (double)(rmax - rmin) * 10.0, instr_per_simd = (double)iters * per * wps;
        printf("  %dw: %5.2f cyc @%.2f GHz", wps, wall_ns * clk / instr_per_simd, clk);
    }
    printf("   (shader cycles per wave-instruction per SIMD, whole-launch wall)\n");
}
int main()
{
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 256 * 2048 * 4 * 2); (void)hipMalloc(&cyc, 256 * 8 * 4 * 8 * 4);
    run<0>("v_fmac v,v,v  8 chains", out, cyc); run<1>("v_fmac v,S,v  8 chains", out, cyc); run<2>("v_mul v,S,v  8 chains", out, cyc);
    run<3>("v_fmac acc,S,ring  5 chains 16 ring regs", out, cyc); run<4>("v_fmac acc,v,ring  5 chains 16 ring regs", out, cyc);
    run<5>("v_fma vop3 v,v,v,v  8 chains", out, cyc); run<6>("v_pk_fma_f32 4 chains (2 fma per lane)", out, cyc); run<7>("v_add v,v,v", out, cyc);
    run<8>("v_mul v, 2.0 (inline const), v", out, cyc); run<9>("v_add v, literal32, v", out, cyc); run<10>("v_cndmask v,v,v,vcc", out, cyc);
    run<11>("v_cmp -> vcc", out, cyc); run<12>("v_max/v_min v,v,v", out, cyc); run<13>("v_mov v,v", out, cyc); run<14>("int add/shl/and v,v,v", out, cyc);
    return 0;
}
