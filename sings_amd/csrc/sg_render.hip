// Per-tile alpha-composite forward / backward.
//
// Replaces renderCUDA<3> forward and backward of the rasterizer the reference calls at
// sings/rec/renderer/gs_renderer_single.py:87-95.
//
// Forward : one 256-thread workgroup (4 wave64) per 16x16 tile; each wave owns an 8x8 pixel
//           quadrant (better skip coherence than a 16x4 strip).  The tile's depth-sorted list is
//           staged through LDS in batches of 256 (gathered 48-B projected records), every lane
//           blends front to back with early termination.
// Backward: same tiling, back to front.  The 9 per-(pixel,Gaussian) partials are reduced across
//           the 64 lanes with DPP, across the 4 waves in LDS, and stored ONCE per
//           (tile,Gaussian) as a 48-byte record at the Gaussian-major slot reserved in the
//           forward pass.  No float atomics (memory-side atomics cap at ~1.3 TB/s on MI355X and
//           scattered single-row adds are 17x slower); gradients are bitwise reproducible.
#include "sg_common.h"

#define SG_BATCH 256

__device__ __forceinline__ float sg_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// XCD-aware block -> tile map: consecutive tiles (which share Gaussians) go to one XCD's L2.
__device__ __forceinline__ int sg_tile_of_block(int b, int T)
{
    int chunk = (T + 7) >> 3;
    return (b & 7) * chunk + (b >> 3);
}

__global__ void __launch_bounds__(256)
sg_render_fwd_kernel(int W, int H, int gx, int T, const uint2 *__restrict__ ranges,
                     const uint32_t *__restrict__ point_list, const float4 *__restrict__ recA,
                     const float4 *__restrict__ recB, const float4 *__restrict__ recC,
                     const float *__restrict__ bg, float *__restrict__ out_color,
                     float *__restrict__ final_T, uint32_t *__restrict__ n_contrib)
{
    __shared__ float4 sA[SG_BATCH];
    __shared__ float4 sB[SG_BATCH];
    __shared__ float sC[SG_BATCH];
    const int tile = sg_tile_of_block(blockIdx.x, T);
    if (tile >= T) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int px = (tile % gx) * 16 + (wave & 1) * 8 + (lane & 7);
    const int py = (tile / gx) * 16 + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];
    int toDo = (int)(range.y - range.x);
    const int rounds = (toDo + SG_BATCH - 1) / SG_BATCH;
    bool done = !inside;
    float Tr = 1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
    uint32_t contributor = 0, last = 0;
    for (int i = 0; i < rounds; i++, toDo -= SG_BATCH) {
        if (__syncthreads_count(done) == 256) break;
        uint32_t e = range.x + i * SG_BATCH + tid;
        if (e < range.y) {
            uint32_t gid = point_list[e];
            sA[tid] = recA[gid]; sB[tid] = recB[gid]; sC[tid] = recC[gid].x;
        }
        __syncthreads();
        const int nb = toDo < SG_BATCH ? toDo : SG_BATCH;
        for (int j = 0; !done && j < nb; j++) {
            contributor++;
            float4 a = sA[j], b = sB[j];
            float dx = a.x - pxf, dy = a.y - pyf;
            // power = -0.5 (cx dx^2 + cz dy^2) - cy dx dy
            float power = fmaf(-0.5f, fmaf(a.z * dx, dx, b.x * dy * dy), -(a.w * dx) * dy);
            if (power > 0.0f) continue;
            float alpha = fminf(0.99f, b.y * sg_exp(power));
            if (alpha < 1.0f / 255.0f) continue;
            float test_T = Tr * (1.0f - alpha);
            if (test_T < 0.0001f) { done = true; continue; }
            float w = alpha * Tr;
            C0 = fmaf(b.z, w, C0); C1 = fmaf(b.w, w, C1); C2 = fmaf(sC[j], w, C2);
            Tr = test_T;
            last = contributor;
        }
    }
    if (inside) {
        size_t pid = (size_t)py * W + px, hw = (size_t)H * W;
        final_T[pid] = Tr;
        n_contrib[pid] = last;
        out_color[pid] = fmaf(Tr, bg[0], C0);
        out_color[hw + pid] = fmaf(Tr, bg[1], C1);
        out_color[2 * hw + pid] = fmaf(Tr, bg[2], C2);
    }
}

void sg_launch_render_fwd(const SgCam &c, SgGeom g, SgBin b, size_t cap, SgImg im, float *out_color,
                          hipStream_t st)
{
    (void)cap;
    const int T = c.gx * c.gy;
    const int grid = ((T + 7) / 8) * 8;
    sg_prof_begin(SG_K_RENDER_FWD, st);
    hipLaunchKernelGGL(sg_render_fwd_kernel, dim3(grid), dim3(256), 0, st, c.W, c.H, c.gx, T, b.ranges,
                       b.point_list, g.recA, g.recB, g.recC, c.bg, out_color, im.final_T, im.n_contrib);
    sg_prof_end(SG_K_RENDER_FWD, st);
}

// ------------------------------------------------------------------------------------------
// wave64 sum; the total lands in lane 63
#define SG_DPP_ADD(v, ctrl, rmask)                                                                   \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl,  \
                                                               rmask, 0xF, false))
__device__ __forceinline__ float sg_wave_sum63(float v)
{
    SG_DPP_ADD(v, 0xB1, 0xF);    // quad_perm [1,0,3,2]
    SG_DPP_ADD(v, 0x4E, 0xF);    // quad_perm [2,3,0,1]
    SG_DPP_ADD(v, 0x141, 0xF);   // row_half_mirror
    SG_DPP_ADD(v, 0x140, 0xF);   // row_mirror        -> every lane holds its row's sum
    SG_DPP_ADD(v, 0x142, 0xA);   // row_bcast15 into rows 1,3
    SG_DPP_ADD(v, 0x143, 0xC);   // row_bcast31 into rows 2,3 -> lane 63 = wave total
    return v;
}

__global__ void __launch_bounds__(256)
sg_render_bwd_kernel(int W, int H, int gx, int T, const uint2 *__restrict__ ranges,
                     const uint32_t *__restrict__ point_list, const float4 *__restrict__ recA,
                     const float4 *__restrict__ recB, const float4 *__restrict__ recC,
                     const float *__restrict__ bg, const float *__restrict__ final_T,
                     const uint32_t *__restrict__ n_contrib, const float *__restrict__ dL_dpix,
                     float4 *__restrict__ grec, uint32_t cap)
{
    __shared__ float4 sA[SG_BATCH];
    __shared__ float4 sB[SG_BATCH];
    __shared__ float4 sC[SG_BATCH];
    __shared__ float wbuf[4][SG_BATCH][9];
    __shared__ uint32_t wflag[SG_BATCH];      // byte w of wflag[j] != 0: wave w wrote wbuf[w][j]
    __shared__ uint32_t smax[4];
    const int tile = sg_tile_of_block(blockIdx.x, T);
    if (tile >= T) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tx = tile % gx, ty = tile / gx;
    const int px = tx * 16 + (wave & 1) * 8 + (lane & 7);
    const int py = ty * 16 + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    if (n == 0) return;
    const size_t pid = (size_t)py * W + px, hw = (size_t)H * W;
    const float T_final = inside ? final_T[pid] : 0.0f;
    const uint32_t last_contributor = inside ? n_contrib[pid] : 0u;
    float dLp0 = 0, dLp1 = 0, dLp2 = 0;
    if (inside) { dLp0 = dL_dpix[pid]; dLp1 = dL_dpix[hw + pid]; dLp2 = dL_dpix[2 * hw + pid]; }
    const float bg_dot = bg[0] * dLp0 + bg[1] * dLp1 + bg[2] * dLp2;
    // tile-wide max of n_contrib: entries beyond it were blended by no pixel
    uint32_t m = last_contributor;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { uint32_t u = __shfl_xor(m, o, 64); m = u > m ? u : m; }
    if (lane == 0) smax[wave] = m;
    __syncthreads();
    const int max_contrib = (int)max(max(smax[0], smax[1]), max(smax[2], smax[3]));
    float Tr = T_final, ar0 = 0, ar1 = 0, ar2 = 0, lc0 = 0, lc1 = 0, lc2 = 0, last_alpha = 0;
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
    const int nbatches = (n + SG_BATCH - 1) / SG_BATCH;
    for (int k = nbatches - 1; k >= 0; k--) {
        const int lo = k * SG_BATCH;
        const int cnt = n - lo < SG_BATCH ? n - lo : SG_BATCH;
        const bool mine = tid < cnt;
        float4 myC = make_float4(0, 0, 0, 0);
        __syncthreads();                       // previous batch fully consumed
        if (mine) {
            uint32_t gid = point_list[range.x + lo + tid];
            myC = recC[gid];
            if (lo < max_contrib) { sA[tid] = recA[gid]; sB[tid] = recB[gid]; sC[tid] = myC; }
        }
        wflag[tid] = 0;
        __syncthreads();
        if (lo < max_contrib) {
            const int hi = cnt < max_contrib - lo ? cnt : max_contrib - lo;
            for (int j = hi - 1; j >= 0; j--) {
                const uint32_t e = (uint32_t)(lo + j);
                float v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0, v6 = 0, v7 = 0, v8 = 0;
                bool hit = false;
                if (e < last_contributor) {
                    float4 a = sA[j], b = sB[j];
                    float dx = a.x - pxf, dy = a.y - pyf;
                    float power = fmaf(-0.5f, fmaf(a.z * dx, dx, b.x * dy * dy), -(a.w * dx) * dy);
                    if (!(power > 0.0f)) {
                        float G = sg_exp(power);
                        float alpha = fminf(0.99f, b.y * G);
                        if (!(alpha < 1.0f / 255.0f)) {
                            hit = true;
                            float cb = sC[j].x;
                            Tr = Tr * __builtin_amdgcn_rcpf(1.0f - alpha);
                            float dchan = alpha * Tr;
                            ar0 = fmaf(last_alpha, lc0, (1.0f - last_alpha) * ar0);
                            ar1 = fmaf(last_alpha, lc1, (1.0f - last_alpha) * ar1);
                            ar2 = fmaf(last_alpha, lc2, (1.0f - last_alpha) * ar2);
                            lc0 = b.z; lc1 = b.w; lc2 = cb;
                            float dL_dalpha = (b.z - ar0) * dLp0 + (b.w - ar1) * dLp1 + (cb - ar2) * dLp2;
                            v6 = dchan * dLp0; v7 = dchan * dLp1; v8 = dchan * dLp2;
                            dL_dalpha *= Tr;
                            last_alpha = alpha;
                            dL_dalpha += (-T_final * __builtin_amdgcn_rcpf(1.0f - alpha)) * bg_dot;
                            float dL_dG = b.y * dL_dalpha;
                            float gdx = G * dx, gdy = G * dy;
                            float dG_ddelx = -gdx * a.z - gdy * a.w;
                            float dG_ddely = -gdy * b.x - gdx * a.w;
                            v0 = dL_dG * dG_ddelx * ddelx_dx;
                            v1 = dL_dG * dG_ddely * ddely_dy;
                            v2 = -0.5f * gdx * dx * dL_dG;
                            v3 = -0.5f * gdx * dy * dL_dG;
                            v4 = -0.5f * gdy * dy * dL_dG;
                            v5 = G * dL_dalpha;
                        }
                    }
                }
                if (__ballot(hit) == 0ull) continue;          // wave-uniform
                v0 = sg_wave_sum63(v0); v1 = sg_wave_sum63(v1); v2 = sg_wave_sum63(v2);
                v3 = sg_wave_sum63(v3); v4 = sg_wave_sum63(v4); v5 = sg_wave_sum63(v5);
                v6 = sg_wave_sum63(v6); v7 = sg_wave_sum63(v7); v8 = sg_wave_sum63(v8);
                if (lane == 63) {
                    float *o = wbuf[wave][j];
                    o[0] = v0; o[1] = v1; o[2] = v2; o[3] = v3; o[4] = v4; o[5] = v5; o[6] = v6; o[7] = v7; o[8] = v8;
                    ((volatile uint8_t *)&wflag[j])[wave] = 1;
                }
            }
        }
        __syncthreads();
        if (mine) {
            float s[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
            uint32_t f = wflag[tid];
#pragma unroll
            for (int w = 0; w < 4; w++)
                if ((f >> (8 * w)) & 0xffu) {
#pragma unroll
                    for (int q = 0; q < 9; q++) s[q] += wbuf[w][tid][q];
                }
            uint32_t goff = __float_as_uint(myC.y), mn = __float_as_uint(myC.z), wh = __float_as_uint(myC.w);
            int x0 = mn & 0xffff, y0 = mn >> 16, rw = wh & 0xffff;
            size_t r = (size_t)goff + (size_t)((ty - y0) * rw + (tx - x0));
            if (r < cap) {
                grec[3 * r] = make_float4(s[0], s[1], s[2], s[3]);
                grec[3 * r + 1] = make_float4(s[4], s[5], s[6], s[7]);
                grec[3 * r + 2] = make_float4(s[8], 0.0f, 0.0f, 0.0f);
            }
        }
    }
}

void sg_launch_render_bwd(const SgCam &c, SgGeom g, SgBin b, size_t cap, SgImg im,
                          const float *dL_dpix, float *grec, hipStream_t st)
{
    const int T = c.gx * c.gy;
    const int grid = ((T + 7) / 8) * 8;
    uint32_t cap32 = cap > 0xffffffffull ? 0xffffffffu : (uint32_t)cap;
    sg_prof_begin(SG_K_RENDER_BWD, st);
    hipLaunchKernelGGL(sg_render_bwd_kernel, dim3(grid), dim3(256), 0, st, c.W, c.H, c.gx, T, b.ranges,
                       b.point_list, g.recA, g.recB, g.recC, c.bg, im.final_T, im.n_contrib, dL_dpix,
                       (float4 *)grec, cap32);
    sg_prof_end(SG_K_RENDER_BWD, st);
}
