// Per-tile alpha-composite forward / backward.
//
// Replaces renderCUDA<3> forward and backward of the rasterizer the reference calls at
// sings/rec/renderer/gs_renderer_single.py:87-95.
//
// Both kernels are VALU-issue bound (not HBM): ~2e8 pixel x Gaussian evaluations per direction at
// cfg3.  wave64 design:
//  * ONE wave per 16x16 tile (4 independent tiles per 256-thread workgroup, no workgroup barriers);
//    each lane owns 4 pixels, one in each 8x8 quadrant of the tile.
//  * The tile's depth-sorted list is staged through LDS 64 entries at a time (three aligned 16-B
//    gathers per entry).  While staging, each lane computes for ITS entry which quadrants the
//    alpha >= 1/255 ellipse can reach (conservative bounding box from conic + opacity); entries that
//    reach nothing are dropped by a ballot compaction, and a quadrant pass runs only where the bit is
//    set -> ~2.5x fewer wave-level evaluations than the upstream 16x16 block, identical results.
//  * Backward: the 9 per-pixel partials are first summed over the lane's quadrant passes in
//    registers, then across the 64 lanes with a multi-value butterfly (v_permlane32_swap /
//    v_permlane16_swap / DPP: 24 ops for 9 values instead of 54), and stored ONCE per
//    (tile,Gaussian) as a 48-byte record at the Gaussian-major slot reserved in the forward pass.
//    No float atomics (memory-side atomics cap at ~1.3 TB/s on MI355X and scattered single-row adds
//    are 17x slower); gradients are bitwise reproducible.
#include "sg_common.h"

#define SG_WB 64          // list entries staged per batch (one per lane)

__device__ __forceinline__ float sg_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// XCD-aware map: workgroup b runs on XCD b % 8 (round-robin dispatch); give each XCD a contiguous
// range of tiles so that neighbouring tiles (which share Gaussians) hit the same L2.
__device__ __forceinline__ int sg_tile_of_wave(int block, int wave, int nblocks)
{
    int chunk = nblocks >> 3;                     // nblocks is a multiple of 8
    return (((block & 7) * chunk + (block >> 3)) << 2) + wave;
}

// Which 8x8 quadrants of the tile at (X0,Y0) can this entry reach with alpha >= 1/255?
// alpha = min(.99, o exp(power)) >= 1/255  =>  power >= -tau, tau = ln(255 o), i.e. the pixel lies in the
// ellipse q(d) = A dx^2 + 2 B dx dy + C dy^2 <= 2 tau (d = pixel - mean).  A quadrant is kept iff the
// minimum of q over its pixel rectangle (0 if the mean is inside, else the minimum over the four edges,
// each a clamped 1-D parabola) is within the bound.  tau and the comparison carry safety margins far
// above the fp32 rounding of power / exp / this test, so a pixel the blend loop would accept is never
// dropped; ill-conditioned conics keep all quadrants.
__device__ __forceinline__ float sg_min_q_rect(float A, float B, float C, float mx, float my, float x0, float x1,
                                               float y0, float y1)
{
    if (mx >= x0 && mx <= x1 && my >= y0 && my <= y1) return 0.0f;
    const float rA = __builtin_amdgcn_rcpf(A), rC = __builtin_amdgcn_rcpf(C);
    float best = 3.0e38f;
#pragma unroll
    for (int e = 0; e < 2; e++) {
        float dx = (e ? x1 : x0) - mx;
        float dy = fminf(fmaxf(-B * dx * rC, y0 - my), y1 - my);
        best = fminf(best, fmaf(A * dx, dx, fmaf(2.0f * B * dx, dy, C * dy * dy)));
        float dy2 = (e ? y1 : y0) - my;
        float dx2 = fminf(fmaxf(-B * dy2 * rA, x0 - mx), x1 - mx);
        best = fminf(best, fmaf(A * dx2, dx2, fmaf(2.0f * B * dx2, dy2, C * dy2 * dy2)));
    }
    return best;
}

__device__ __forceinline__ uint32_t sg_quad_mask(float4 a, float4 b, float X0, float Y0)
{
    const float o255 = 255.0f * b.y;
    if (!(o255 >= 0.999f)) return 0u;
    const float bound = 2.0f * (__logf(o255) * 1.002f + 0.004f) + 0.01f;
    const float A = a.z, B = a.w, C = b.x;
    const float det = A * C - B * B;
    if (!(det > 1e-3f * A * C) || !(A > 0.0f) || !(C > 0.0f)) return 0xFu;
    const float mx = a.x - X0, my = a.y - Y0;
    uint32_t m = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float x0 = 8.0f * (q & 1), y0 = 8.0f * (q >> 1);
        float mq = sg_min_q_rect(A, B, C, mx, my, x0, x0 + 7.0f, y0, y0 + 7.0f);
        if (mq * 0.999f <= bound) m |= 1u << q;
    }
    return m;
}

struct SgWaveLds {
    float4 sA[SG_WB];
    float4 sB[SG_WB];
    float sC[SG_WB];
    uint32_t sM[SG_WB];     // (entry index << 4) | quadrant mask
    uint32_t sR[SG_WB];     // backward: gradient-record slot
    float sG[SG_WB][9];     // backward: reduced partials
};

__global__ void __launch_bounds__(256)
sg_render_fwd_kernel(int W, int H, int gx, int T, int nblocks, const uint2 *__restrict__ ranges,
                     const uint32_t *__restrict__ point_list, const float4 *__restrict__ recA,
                     const float4 *__restrict__ recB, const float4 *__restrict__ recC,
                     const float *__restrict__ bg, float *__restrict__ out_color,
                     float *__restrict__ final_T, uint32_t *__restrict__ n_contrib)
{
    __shared__ SgWaveLds lds_all[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = sg_tile_of_wave(blockIdx.x, wave, nblocks);
    if (tile >= T) return;
    SgWaveLds &L = lds_all[wave];
    const int X0 = (tile % gx) * 16, Y0 = (tile / gx) * 16;
    const int lx = lane & 7, ly = lane >> 3;
    const float pxf[2] = { (float)(X0 + lx), (float)(X0 + 8 + lx) };
    const float pyf[2] = { (float)(Y0 + ly), (float)(Y0 + 8 + ly) };
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    float Tq[4], C0[4], C1[4], C2[4];
    uint32_t lastq[4];
    uint32_t dmask = 0;                       // bit q: this lane's pixel in quadrant q is finished
#pragma unroll
    for (int q = 0; q < 4; q++) {
        Tq[q] = 1.0f; C0[q] = C1[q] = C2[q] = 0.0f; lastq[q] = 0;
        if (!(X0 + 8 * (q & 1) + lx < W && Y0 + 8 * (q >> 1) + ly < H)) dmask |= 1u << q;
    }
    const uint32_t outside = dmask;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int base = 0; base < n; base += SG_WB) {
        uint32_t qdone = 0;
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (__ballot((dmask >> q) & 1u) == ~0ull) qdone |= 1u << q;
        if (qdone == 0xFu) break;
        const int e = base + lane;
        uint32_t m = 0;
        float4 a, b; float cb = 0;
        if (e < n) {
            uint32_t gid = point_list[range.x + e];
            a = recA[gid]; b = recB[gid]; cb = recC[gid].x;
            m = sg_quad_mask(a, b, (float)X0, (float)Y0) & ~qdone;
        }
        const unsigned long long bal = __ballot(m != 0u);
        const int cnt = __popcll(bal);
        if (m) {
            int pos = __popcll(bal & lt);
            L.sA[pos] = a; L.sB[pos] = b; L.sC[pos] = cb; L.sM[pos] = ((uint32_t)e << 4) | m;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): this wave's LDS writes have landed
        __builtin_amdgcn_wave_barrier();
        for (int k = 0; k < cnt; k++) {
            const uint32_t mm = __builtin_amdgcn_readfirstlane(L.sM[k]);
            const float4 ga = L.sA[k], gb = L.sB[k];
            const float gc = L.sC[k];
            const uint32_t contributor = (mm >> 4) + 1u;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (!(mm & (1u << q))) continue;                 // wave-uniform (scalar branch)
                // straight-line, predicated: no exec-mask branches inside a quadrant pass
                const float dx = ga.x - pxf[q & 1], dy = ga.y - pyf[q >> 1];
                const float power = fmaf(-0.5f, fmaf(ga.z * dx, dx, gb.x * dy * dy), -(ga.w * dx) * dy);
                const float alpha = fminf(0.99f, gb.y * sg_exp(power));
                const float test_T = Tq[q] * (1.0f - alpha);
                const bool valid = !((dmask >> q) & 1u) & !(power > 0.0f) & !(alpha < 1.0f / 255.0f);
                const bool term = valid & (test_T < 0.0001f);
                const bool blend = valid & !term;
                const float w = blend ? alpha * Tq[q] : 0.0f;
                C0[q] = fmaf(gb.z, w, C0[q]); C1[q] = fmaf(gb.w, w, C1[q]); C2[q] = fmaf(gc, w, C2[q]);
                Tq[q] = blend ? test_T : Tq[q];
                lastq[q] = blend ? contributor : lastq[q];
                dmask |= term ? (1u << q) : 0u;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    const size_t hw = (size_t)H * W;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if ((outside >> q) & 1u) continue;
        size_t pid = (size_t)(Y0 + 8 * (q >> 1) + ly) * W + (X0 + 8 * (q & 1) + lx);
        final_T[pid] = Tq[q];
        n_contrib[pid] = lastq[q];
        out_color[pid] = fmaf(Tq[q], bg0, C0[q]);
        out_color[hw + pid] = fmaf(Tq[q], bg1, C1[q]);
        out_color[2 * hw + pid] = fmaf(Tq[q], bg2, C2[q]);
    }
}

static inline int sg_render_blocks(int T) { return (((T + 3) / 4 + 7) / 8) * 8; }

void sg_launch_render_fwd(const SgCam &c, SgGeom g, SgBin b, size_t cap, SgImg im, float *out_color,
                          hipStream_t st)
{
    (void)cap;
    const int T = c.gx * c.gy;
    const int grid = sg_render_blocks(T);
    sg_prof_begin(SG_K_RENDER_FWD, st);
    hipLaunchKernelGGL(sg_render_fwd_kernel, dim3(grid), dim3(256), 0, st, c.W, c.H, c.gx, T, grid, b.ranges,
                       b.point_list, g.recA, g.recB, g.recC, c.bg, out_color, im.final_T, im.n_contrib);
    sg_prof_end(SG_K_RENDER_FWD, st);
}

// ------------------------------------------------------------------------------------------
#define SG_DPP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, false))

// lanes i and i+32 : returns (x_lo + x_hi | y_lo + y_hi)
__device__ __forceinline__ float sg_fold32(float x, float y)
{
    // inline asm: hipcc (ROCm 7.2) folds "r[0] + r[1]" of __builtin_amdgcn_permlane32_swap into r[0] + r[0]
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    return x + y;
}
// lanes i and i+16 inside each half: rows become (x_r0+x_r1, y_r0+y_r1, x_r2+x_r3, y_r2+y_r3)
__device__ __forceinline__ float sg_fold16(float x, float y)
{
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    return x + y;
}

// Sums v[0..8] over the 64 lanes (fixed order).  Afterwards every lane of the 8-lane group
// g = lane >> 3 holds the total of value SG_RED_IDX(g); lane 63 additionally returns the total of v[8]
// in *v8tot.
__device__ __forceinline__ float sg_reduce9(const float v[9], int lane, float *v8tot)
{
    float s01 = sg_fold32(v[0], v[1]), s23 = sg_fold32(v[2], v[3]);
    float s45 = sg_fold32(v[4], v[5]), s67 = sg_fold32(v[6], v[7]);
    float t0 = sg_fold16(s01, s23);          // rows: v0, v2, v1, v3
    float t1 = sg_fold16(s45, s67);          // rows: v4, v6, v5, v7
    float u = t0 + SG_DPP(t0, 0x128);        // row_ror:8 -> lanes i, i^8 summed
    float w = t1 + SG_DPP(t1, 0x128);
    float z = (lane & 8) ? w : u;            // per row: lanes 0-7 <- first value, 8-15 <- second
    z += SG_DPP(z, 0xB1);                    // quad_perm [1,0,3,2]
    z += SG_DPP(z, 0x4E);                    // quad_perm [2,3,0,1]
    z += SG_DPP(z, 0x141);                   // row_half_mirror
    float x = v[8];
    x += SG_DPP(x, 0xB1); x += SG_DPP(x, 0x4E); x += SG_DPP(x, 0x141); x += SG_DPP(x, 0x140);
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x142, 0xA, 0xF, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x143, 0xC, 0xF, false));
    *v8tot = x;
    return z;
}
// value index held by 8-lane group g = lane >> 3 after sg_reduce9: rows (v0|v4, v2|v6, v1|v5, v3|v7)
__device__ __forceinline__ int sg_red_idx(int lane)
{
    const int row = lane >> 4, half = (lane >> 3) & 1;
    const int base = row == 0 ? 0 : (row == 1 ? 2 : (row == 2 ? 1 : 3));
    return base + 4 * half;
}

__global__ void __launch_bounds__(256)
sg_render_bwd_kernel(int W, int H, int gx, int T, int nblocks, const uint2 *__restrict__ ranges,
                     const uint32_t *__restrict__ point_list, const float4 *__restrict__ recA,
                     const float4 *__restrict__ recB, const float4 *__restrict__ recC,
                     const float *__restrict__ bg, const float *__restrict__ final_T,
                     const uint32_t *__restrict__ n_contrib, const float *__restrict__ dL_dpix,
                     float4 *__restrict__ grec, uint32_t cap)
{
    __shared__ SgWaveLds lds_all[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile = sg_tile_of_wave(blockIdx.x, wave, nblocks);
    if (tile >= T) return;
    SgWaveLds &L = lds_all[wave];
    const int tx = tile % gx, ty = tile / gx;
    const int X0 = tx * 16, Y0 = ty * 16;
    const int lx = lane & 7, ly = lane >> 3;
    const float pxf[2] = { (float)(X0 + lx), (float)(X0 + 8 + lx) };
    const float pyf[2] = { (float)(Y0 + ly), (float)(Y0 + 8 + ly) };
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    if (n == 0) return;
    const size_t hw = (size_t)H * W;
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    // per-pixel state (x4 quadrants): running T, colour accumulated BEHIND the current entry (S), dL/dpixel,
    // T_final * <bg, dL/dpixel>, n_contrib
    float Tr[4], S0[4], S1[4], S2[4], d0[4], d1[4], d2[4], tb[4];
    uint32_t ncq[4];
    uint32_t maxc[4];
    int max_contrib = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int px = X0 + 8 * (q & 1) + lx, py = Y0 + 8 * (q >> 1) + ly;
        const bool inside = px < W && py < H;
        const size_t pid = (size_t)py * W + px;
        const float Tfin = inside ? final_T[pid] : 0.0f;
        ncq[q] = inside ? n_contrib[pid] : 0u;
        d0[q] = inside ? dL_dpix[pid] : 0.0f; d1[q] = inside ? dL_dpix[hw + pid] : 0.0f; d2[q] = inside ? dL_dpix[2 * hw + pid] : 0.0f;
        tb[q] = Tfin * (bg0 * d0[q] + bg1 * d1[q] + bg2 * d2[q]);
        Tr[q] = Tfin; S0[q] = S1[q] = S2[q] = 0.0f;
        uint32_t m = ncq[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { uint32_t u = __shfl_xor(m, o, 64); m = u > m ? u : m; }
        maxc[q] = __builtin_amdgcn_readfirstlane(m);
        max_contrib = max_contrib > (int)maxc[q] ? max_contrib : (int)maxc[q];
    }
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int nbatches = (n + SG_WB - 1) / SG_WB;
    const int ridx = sg_red_idx(lane);
    for (int kb = nbatches - 1; kb >= 0; kb--) {
        const int e = kb * SG_WB + lane;
        uint32_t m = 0, rslot = 0xffffffffu;
        float4 a, b, c4;
        if (e < n) {
            uint32_t gid = point_list[range.x + e];
            c4 = recC[gid];
            uint32_t goff = __float_as_uint(c4.y), mn = __float_as_uint(c4.z), wh = __float_as_uint(c4.w);
            int x0 = mn & 0xffff, y0 = mn >> 16, rw = wh & 0xffff;
            rslot = goff + (uint32_t)((ty - y0) * rw + (tx - x0));
            if (e < max_contrib) {
                a = recA[gid]; b = recB[gid];
                m = sg_quad_mask(a, b, (float)X0, (float)Y0);
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (!((uint32_t)e < maxc[q])) m &= ~(1u << q);
            }
            if (m == 0u && rslot < cap) {          // reaches no pixel of this tile: zero record
                float4 z4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                grec[3 * (size_t)rslot] = z4; grec[3 * (size_t)rslot + 1] = z4; grec[3 * (size_t)rslot + 2] = z4;
            }
        }
        const unsigned long long bal = __ballot(m != 0u);
        const int cnt = __popcll(bal);
        if (cnt == 0) continue;
        if (m) {
            int pos = __popcll(bal & lt);
            L.sA[pos] = a; L.sB[pos] = b; L.sC[pos] = c4.x; L.sM[pos] = ((uint32_t)e << 4) | m; L.sR[pos] = rslot;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        for (int k = cnt - 1; k >= 0; k--) {
            const uint32_t mm = __builtin_amdgcn_readfirstlane(L.sM[k]);
            const float4 ga = L.sA[k], gb = L.sB[k];
            const float gc = L.sC[k];
            const uint32_t ee = mm >> 4;
            // v[0..4]: sums of w*m with w = G dL/dalpha; the per-Gaussian factors (-o W/2, -o H/2, -o/2)
            // are applied once per record when it is flushed (the reduction is linear).
            float v[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (!(mm & (1u << q))) continue;                 // wave-uniform (scalar branch)
                // straight-line, predicated (alpha_eff = 0 makes every update an exact no-op)
                const float dx = ga.x - pxf[q & 1], dy = ga.y - pyf[q >> 1];
                const float power = fmaf(-0.5f, fmaf(ga.z * dx, dx, gb.x * dy * dy), -(ga.w * dx) * dy);
                const float G = sg_exp(power);
                const float alpha = fminf(0.99f, gb.y * G);
                const bool valid = (ee < ncq[q]) & !(power > 0.0f) & !(alpha < 1.0f / 255.0f);
                const float ae = valid ? alpha : 0.0f;
                const float rinv = __builtin_amdgcn_rcpf(1.0f - ae);      // rcp(1) == 1 exactly
                Tr[q] = Tr[q] * rinv;                            // T in front of this entry
                const float dchan = ae * Tr[q];
                const float e0 = gb.z - S0[q], e1 = gb.w - S1[q], e2 = gc - S2[q];
                float dLa = fmaf(e2, d2[q], fmaf(e1, d1[q], e0 * d0[q]));
                S0[q] = fmaf(ae, e0, S0[q]); S1[q] = fmaf(ae, e1, S1[q]); S2[q] = fmaf(ae, e2, S2[q]);
                v[6] = fmaf(dchan, d0[q], v[6]); v[7] = fmaf(dchan, d1[q], v[7]); v[8] = fmaf(dchan, d2[q], v[8]);
                dLa = fmaf(-tb[q], rinv, dLa * Tr[q]);          // + (-T_final / (1 - alpha)) <bg, dL/dpixel>
                const float w = valid ? G * dLa : 0.0f;          // = dL/dopacity contribution; dL/dG = o * dLa
                v[5] += w;
                v[0] = fmaf(w, fmaf(dy, ga.w, dx * ga.z), v[0]);
                v[1] = fmaf(w, fmaf(dx, ga.w, dy * gb.x), v[1]);
                const float wx = w * dx;
                v[2] = fmaf(wx, dx, v[2]); v[3] = fmaf(wx, dy, v[3]); v[4] = fmaf(w * dy, dy, v[4]);
            }
            // (entries that pass the quadrant test but touch no pixel are < 1 %: always reduce)
            float v8;
            float z = sg_reduce9(v, lane, &v8);
            if ((lane & 7) == 0) L.sG[k][ridx] = z;
            if (lane == 63) L.sG[k][8] = v8;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        if (lane < cnt) {
            uint32_t r = L.sR[lane];
            if (r < cap) {
                const float *s = L.sG[lane];
                const float no = -L.sB[lane].y, nh = 0.5f * no;          // -opacity, -opacity / 2
                grec[3 * (size_t)r] = make_float4(no * ddelx_dx * s[0], no * ddely_dy * s[1], nh * s[2], nh * s[3]);
                grec[3 * (size_t)r + 1] = make_float4(nh * s[4], s[5], s[6], s[7]);
                grec[3 * (size_t)r + 2] = make_float4(s[8], 0.0f, 0.0f, 0.0f);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

void sg_launch_render_bwd(const SgCam &c, SgGeom g, SgBin b, size_t cap, SgImg im,
                          const float *dL_dpix, float *grec, hipStream_t st)
{
    const int T = c.gx * c.gy;
    const int grid = sg_render_blocks(T);
    uint32_t cap32 = cap > 0xffffffffull ? 0xffffffffu : (uint32_t)cap;
    sg_prof_begin(SG_K_RENDER_BWD, st);
    hipLaunchKernelGGL(sg_render_bwd_kernel, dim3(grid), dim3(256), 0, st, c.W, c.H, c.gx, T, grid, b.ranges,
                       b.point_list, g.recA, g.recB, g.recC, c.bg, im.final_T, im.n_contrib, dL_dpix,
                       (float4 *)grec, cap32);
    sg_prof_end(SG_K_RENDER_BWD, st);
}
