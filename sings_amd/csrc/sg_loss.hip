// Photometric loss of one rendered view, forward + gradient in one pass over the image.
//
// Replaces, for the L1 and SSIM terms of HumanSceneLoss.forward (sings/rec/losses/loss.py:55-69), the chain
//   torch.clamp(rendered, 0, 1)                                  gs_renderer_single.py:96
//   gt = rgb * mask + bg * (1 - mask)                             loss.py:58
//   l1_w * |pred - gt|.sum() / mask.sum()                         losses/utils.py:16-20
//   ssim_w * (1 - ssim(pred, gt)) * mask.sum() / (H W)            losses/utils.py:28-70, loss.py:65-67
// and their autograd backward down to dL/d(rasterizer output).  (The LPIPS term is a VGG network: out of scope.)
//
// The reference runs ~25 elementwise / conv2d kernels each way; here:
//   pass 1  sg_ssim_stats_kernel : clamp + composite on load, 11x11 Gaussian window as two separable passes in LDS,
//                                  SSIM map and its three partial-derivative maps, per-workgroup partial sums
//   reduce  sg_loss_reduce_kernel: fixed-order sum of the partials -> the four loss scalars and the gradient scales
//   pass 2  sg_ssim_grad_kernel  : the same separable window over the three derivative maps (a zero-padded
//                                  symmetric window is its own adjoint), + the L1 sign term, x the clamp mask
// HBM-bound streaming: algorithmic bytes per pixel = 3 ch x (raw 4 + gt 4 + grad 4) + mask 4 = 40; implementation
// traffic adds the three fp32 derivative maps written and read once (72 B/pixel).  No float atomics: deterministic.
#include "sg_common.h"

#define SG_LT 32                  // output tile edge
#define SG_LH (SG_LT + 10)        // with the 5-pixel halo of the 11-tap window
#define SG_LP (SG_LH + 1)         // LDS row pitch (bank-conflict padding)
#define SG_LOADS ((SG_LH * SG_LH + 255) / 256)     // halo pixels per thread

struct SgLossArgs {
    int W, H;
    float l1_w, ssim_w;
    float w[11];                  // the reference's fp32 window: exp(-(x-5)^2 / (2 * 1.5^2)), normalised
    // K frames per launch (round 4): blockIdx.z = 3 frame + channel.  Frame f: raw / gradient / optional images at + 3 H W f,
    // target at + gt_stride f, mask at + mask_stride f (floats; 0 = one target / mask for all frames), workspace at + ws_stride f
    // bytes, losses at + 4 f
    size_t gt_stride, mask_stride, ws_stride;
    int up_stride;                // floats between the upstream weight pairs of consecutive frames: 0 (shared) | 2
};

__device__ __forceinline__ float sg_clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

// true for all threads iff every thread's `same` holds and (v0, v1, v2) have the same bits in all 256 threads; contains the
// workgroup barrier that also publishes the tile the caller has just stored to LDS.
__device__ __forceinline__ bool sg_tile_is_flat(bool same, float v0, float v1, float v2, uint32_t (*sFlat)[4], int lane, int wave)
{
    const uint32_t b0 = __float_as_uint(v0), b1 = __float_as_uint(v1), b2 = __float_as_uint(v2);
    const uint32_t f0 = __builtin_amdgcn_readfirstlane(b0), f1 = __builtin_amdgcn_readfirstlane(b1), f2 = __builtin_amdgcn_readfirstlane(b2);
    const bool w = __all(same && b0 == f0 && b1 == f1 && b2 == f2);
    if (lane == 0) { sFlat[wave][0] = w ? 1u : 0u; sFlat[wave][1] = f0; sFlat[wave][2] = f1; sFlat[wave][3] = f2; }
    __syncthreads();
    bool flat = true;
#pragma unroll
    for (int k = 0; k < 4; k++)
        flat = flat && sFlat[k][0] != 0u && sFlat[k][1] == sFlat[0][1] && sFlat[k][2] == sFlat[0][2] && sFlat[k][3] == sFlat[0][3];
    return flat;
}

// block partial layout: [block][4] = (sum |pred - gt|, sum ssim, sum mask, unused)
__global__ void __launch_bounds__(256)
sg_ssim_stats_kernel(SgLossArgs a, const float *__restrict__ raw, const float *__restrict__ gt_rgb,
                     const float *__restrict__ mask, const float *__restrict__ bg, float *__restrict__ maps,
                     float *__restrict__ pred_out, float *__restrict__ gt_out, float4 *__restrict__ partial)
{
    __shared__ float sX[SG_LH][SG_LP], sY[SG_LH][SG_LP];
    __shared__ float sH[SG_LH][SG_LT + 1];
    __shared__ float sRed[4][3];
    __shared__ uint32_t sFlat[4][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int X0 = blockIdx.x * SG_LT, Y0 = blockIdx.y * SG_LT;
    const size_t hw = (size_t)a.W * a.H;
    float acc_l1 = 0.0f, acc_ssim = 0.0f, acc_mask = 0.0f;
    const int frame = blockIdx.z / 3;
    {
        const size_t fi = (size_t)frame * 3 * hw;
        raw += fi; gt_rgb += (size_t)frame * a.gt_stride; mask += (size_t)frame * a.mask_stride;
        maps = sg_at(maps, (size_t)frame * a.ws_stride); partial = sg_at(partial, (size_t)frame * a.ws_stride);
        if (pred_out) pred_out += fi;
        if (gt_out) gt_out += fi;
    }
    {   // one (tile, channel) per workgroup: a 512x896 frame is only 448 tiles, fewer than two per CU
        const int ch = blockIdx.z - 3 * frame;
        const float bgc = bg[ch];
        // the halo tile: all 3 x SG_LOADS loads of a thread are in flight together (as a rolled loop hipcc emitted load, load, load,
        // s_waitcnt vmcnt(0) per trip -- seven dependent memory round trips in front of the first barrier of a workgroup whose
        // arithmetic is ~800 instructions per wave).  Addresses are clamped into the image instead of branched around.
        bool same = true;
        float x0 = 0.0f, y0 = 0.0f;
        {
            float lx[SG_LOADS], ly[SG_LOADS], lm[SG_LOADS];
#pragma unroll
            for (int u = 0; u < SG_LOADS; u++) {
                const int i = tid + 256 * u, r = i / SG_LH, c = i - r * SG_LH;
                const int x = X0 - 5 + c, y = Y0 - 5 + r;
                const int xc = x < 0 ? 0 : (x < a.W ? x : a.W - 1), yc = y < 0 ? 0 : (y < a.H ? y : a.H - 1);
                const size_t p = (size_t)yc * a.W + xc;
                lm[u] = mask[p]; lx[u] = raw[ch * hw + p]; ly[u] = gt_rgb[ch * hw + p];
            }
#pragma unroll
            for (int u = 0; u < SG_LOADS; u++) {
                const int i = tid + 256 * u, r = i / SG_LH, c = i - r * SG_LH;
                const int x = X0 - 5 + c, y = Y0 - 5 + r;
                const bool in = x >= 0 && x < a.W && y >= 0 && y < a.H;
                const float m = lm[u];
                const float xv = in ? sg_clamp01(lx[u]) : 0.0f, yv = in ? ly[u] * m + bgc * (1.0f - m) : 0.0f;
                if (i < SG_LH * SG_LH) { sX[r][c] = xv; sY[r][c] = yv; }
                if (u == 0) { x0 = xv; y0 = yv; }
                else if (i < SG_LH * SG_LH) same = same && __float_as_uint(xv) == __float_as_uint(x0) && __float_as_uint(yv) == __float_as_uint(y0);
            }
        }
        const bool flat = sg_tile_is_flat(same, x0, y0, 0.0f, sFlat, lane, wave);        // (holds the barrier behind the LDS stores)
        // Separable window, one quantity at a time through ONE LDS plane (20 KB per workgroup instead of 42 KB: the
        // kernel is latency-bound, resident workgroups are what hides it).  Horizontal pass: an item = 8 consecutive
        // output columns of one halo row, its 18 samples of x and y stay in registers for all five quantities;
        // vertical pass: 4 consecutive rows of one column per thread (14 LDS reads for 4 outputs).
        const bool hthread = tid < SG_LH * 4;
        const int hr = tid >> 2, hc0 = (tid & 3) * 8;
        float xs[18], ys[18];
        if (hthread && !flat) {
#pragma unroll
            for (int k = 0; k < 18; k++) { xs[k] = sX[hr][hc0 + k]; ys[k] = sY[hr][hc0 + k]; }
        }
        const int c = tid & 31, r0 = (tid >> 5) * 4;
        float vq[5][4];
        if (flat) {
            // every sample of the halo tile has the same bits (background behind the avatar: render = bg, target = bg; or the zero
            // padding outside the image): each of the 8 x 11 and 4 x 11 FMA chains below would run on the same operands in the same
            // order, so ONE chain per quantity IS the value of all of them -- no LDS traffic, no barriers, identical bits
#pragma unroll
            for (int q = 0; q < 5; q++) {
                const float v = q == 0 ? x0 : q == 1 ? y0 : q == 2 ? x0 * x0 : q == 3 ? y0 * y0 : x0 * y0;
                float h = 0.0f, t = 0.0f;
#pragma unroll
                for (int k = 0; k < 11; k++) h = fmaf(a.w[k], v, h);
#pragma unroll
                for (int k = 0; k < 11; k++) t = fmaf(a.w[k], h, t);
#pragma unroll
                for (int j = 0; j < 4; j++) vq[q][j] = t;
            }
        } else
#pragma unroll
        for (int q = 0; q < 5; q++) {
            if (hthread) {
                float v[18];
#pragma unroll
                for (int k = 0; k < 18; k++)
                    v[k] = q == 0 ? xs[k] : q == 1 ? ys[k] : q == 2 ? xs[k] * xs[k] : q == 3 ? ys[k] * ys[k] : xs[k] * ys[k];
#pragma unroll
                for (int o = 0; o < 8; o++) {
                    float h = 0.0f;
#pragma unroll
                    for (int k = 0; k < 11; k++) h = fmaf(a.w[k], v[o + k], h);
                    sH[hr][hc0 + o] = h;
                }
            }
            __syncthreads();
            float col[14];
#pragma unroll
            for (int k = 0; k < 14; k++) col[k] = sH[r0 + k][c];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float h = 0.0f;
#pragma unroll
                for (int k = 0; k < 11; k++) h = fmaf(a.w[k], col[j + k], h);
                vq[q][j] = h;
            }
            __syncthreads();
        }
        float mk[4] = { 0.0f, 0.0f, 0.0f, 0.0f };           // (channel 0 also sums the mask: its four loads go out together)
        if (ch == 0) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int x = X0 + c, y = Y0 + r0 + j;
                mk[j] = mask[(size_t)(y < a.H ? y : a.H - 1) * a.W + (x < a.W ? x : a.W - 1)];       // (clamped, not branched: used in-image only)
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int r = r0 + j;
            const int x = X0 + c, y = Y0 + r;
            const float mu1 = vq[0][j], mu2 = vq[1][j], e11 = vq[2][j], e22 = vq[3][j], e12 = vq[4][j];
            if (x < a.W && y < a.H) {
                const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
                const float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
                const float s1 = e11 - mu1s, s2 = e22 - mu2s, s12 = e12 - mu12;
                const float A = mu1s + mu2s + C1, B = s1 + s2 + C2, Cn = 2.0f * mu12 + C1, D = 2.0f * s12 + C2;
                const float rA = 1.0f / A, rB = 1.0f / B;
                const float m = (Cn * D) * (rA * rB);
                // partials of m with respect to the window outputs mu1 = w*x, E[x^2] = w*x^2, E[xy] = w*xy
                const float dm_ds1 = -m * rB;                          // d/d sigma1^2
                const float dm_ds12 = 2.0f * Cn * (rA * rB);            // d/d sigma12
                const float dm_dmu1 = 2.0f * mu2 * D * (rA * rB) - 2.0f * mu1 * m * rA
                                      - 2.0f * mu1 * dm_ds1 - mu2 * dm_ds12;
                const size_t p = (size_t)y * a.W + x;
                maps[(size_t)(ch * 3 + 0) * hw + p] = dm_dmu1;
                maps[(size_t)(ch * 3 + 1) * hw + p] = dm_ds1;
                maps[(size_t)(ch * 3 + 2) * hw + p] = dm_ds12;
                const float xv = sX[r + 5][c + 5], yv = sY[r + 5][c + 5];
                acc_ssim += m;
                acc_l1 += fabsf(xv - yv);
                if (ch == 0) acc_mask += mk[j];
                if (pred_out) pred_out[ch * hw + p] = xv;
                if (gt_out) gt_out[ch * hw + p] = yv;
            }
        }
    }
    // fixed-order block reduction
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        acc_l1 += __shfl_xor(acc_l1, o, 64); acc_ssim += __shfl_xor(acc_ssim, o, 64); acc_mask += __shfl_xor(acc_mask, o, 64);
    }
    if (lane == 0) { sRed[wave][0] = acc_l1; sRed[wave][1] = acc_ssim; sRed[wave][2] = acc_mask; }
    __syncthreads();
    if (tid == 0) {
        float t0 = 0, t1 = 0, t2 = 0;
        for (int w = 0; w < 4; w++) { t0 += sRed[w][0]; t1 += sRed[w][1]; t2 += sRed[w][2]; }
        partial[((blockIdx.z - 3 * frame) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = make_float4(t0, t1, t2, 0.0f);
    }
}

// scalars[0..3] = (l1_w * Ll1, ssim_w * Lssim, Ll1, mean ssim); scalars[4..5] = gradient scales (c_l1, c_ssim)
__global__ void __launch_bounds__(256)
sg_loss_reduce_kernel(SgLossArgs a, const float4 *__restrict__ partial, int nblocks, float *__restrict__ scalars,
                      float *__restrict__ losses)
{
    __shared__ double sR[256][3];
    partial = sg_at(partial, (size_t)blockIdx.x * a.ws_stride); scalars = sg_at(scalars, (size_t)blockIdx.x * a.ws_stride);   // frame blockIdx.x
    if (losses) losses += 4 * blockIdx.x;
    double t0 = 0, t1 = 0, t2 = 0;
    for (int i = threadIdx.x; i < nblocks; i += 256) { const float4 p = partial[i]; t0 += p.x; t1 += p.y; t2 += p.z; }
    sR[threadIdx.x][0] = t0; sR[threadIdx.x][1] = t1; sR[threadIdx.x][2] = t2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int q = 0; q < 3; q++) sR[threadIdx.x][q] += sR[threadIdx.x + s][q];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double hw = (double)a.W * a.H;
        const double s_l1 = sR[0][0], s_ssim = sR[0][1], s_mask = sR[0][2];
        const double Ll1 = s_l1 / s_mask, ssim_mean = s_ssim / (3.0 * hw);
        const double Lssim = (1.0 - ssim_mean) * (s_mask / hw);
        scalars[0] = (float)(a.l1_w * Ll1); scalars[1] = (float)(a.ssim_w * Lssim);
        scalars[2] = (float)Ll1; scalars[3] = (float)ssim_mean;
        scalars[4] = (float)(a.l1_w / s_mask);
        scalars[5] = (float)(-(double)a.ssim_w * (s_mask / hw) / (3.0 * hw));
        if (losses) { losses[0] = scalars[0]; losses[1] = scalars[1]; losses[2] = scalars[2]; losses[3] = scalars[3]; }
    }
}

__global__ void __launch_bounds__(256)
sg_ssim_grad_kernel(SgLossArgs a, const float *__restrict__ raw, const float *__restrict__ gt_rgb,
                    const float *__restrict__ mask, const float *__restrict__ bg, const float *__restrict__ maps,
                    const float *__restrict__ scalars, const float *__restrict__ upstream, float *__restrict__ dL_draw)
{
    __shared__ float sM[3][SG_LH][SG_LP];
    __shared__ float sH[SG_LH][SG_LT + 1];
    __shared__ uint32_t sFlat[4][4];
    const int tid = threadIdx.x;
    const int X0 = blockIdx.x * SG_LT, Y0 = blockIdx.y * SG_LT;
    const size_t hw = (size_t)a.W * a.H;
    const int frame = blockIdx.z / 3;
    {
        const size_t fi = (size_t)frame * 3 * hw;
        raw += fi; gt_rgb += (size_t)frame * a.gt_stride; mask += (size_t)frame * a.mask_stride; dL_draw += fi;
        maps = sg_at(maps, (size_t)frame * a.ws_stride); scalars = sg_at(scalars, (size_t)frame * a.ws_stride);
    }
    // upstream = (d loss / d weighted l1 term, d loss / d weighted ssim term); 1, 1 when NULL (the same for all frames)
    const float *up = upstream ? upstream + (size_t)frame * a.up_stride : nullptr;     // (up_stride 2: a pair of weights per frame)
    const float u_l1 = up ? up[0] : 1.0f, u_ss = up ? up[1] : 1.0f;
    const float c_l1 = scalars[4] * u_l1, c_ss = scalars[5] * u_ss;
    {
        const int ch = blockIdx.z - 3 * frame;
        bool same = true;
        float f0 = 0.0f, f1 = 0.0f, f2 = 0.0f;
        {   // (all loads of the halo tile in flight together: see sg_ssim_stats_kernel)
            float l0[SG_LOADS], l1[SG_LOADS], l2[SG_LOADS];
#pragma unroll
            for (int u = 0; u < SG_LOADS; u++) {
                const int i = tid + 256 * u, r = i / SG_LH, c = i - r * SG_LH;
                const int x = X0 - 5 + c, y = Y0 - 5 + r;
                const int xc = x < 0 ? 0 : (x < a.W ? x : a.W - 1), yc = y < 0 ? 0 : (y < a.H ? y : a.H - 1);
                const size_t p = (size_t)yc * a.W + xc;
                l0[u] = maps[(size_t)(ch * 3 + 0) * hw + p]; l1[u] = maps[(size_t)(ch * 3 + 1) * hw + p];
                l2[u] = maps[(size_t)(ch * 3 + 2) * hw + p];
            }
#pragma unroll
            for (int u = 0; u < SG_LOADS; u++) {
                const int i = tid + 256 * u, r = i / SG_LH, c = i - r * SG_LH;
                const int x = X0 - 5 + c, y = Y0 - 5 + r;
                const bool in = x >= 0 && x < a.W && y >= 0 && y < a.H;
                const float m0 = in ? l0[u] : 0.0f, m1 = in ? l1[u] : 0.0f, m2 = in ? l2[u] : 0.0f;
                if (i < SG_LH * SG_LH) { sM[0][r][c] = m0; sM[1][r][c] = m1; sM[2][r][c] = m2; }
                if (u == 0) { f0 = m0; f1 = m1; f2 = m2; }
                else if (i < SG_LH * SG_LH)
                    same = same && __float_as_uint(m0) == __float_as_uint(f0) && __float_as_uint(m1) == __float_as_uint(f1)
                           && __float_as_uint(m2) == __float_as_uint(f2);
            }
        }
        const bool flat = sg_tile_is_flat(same, f0, f1, f2, sFlat, tid & 63, tid >> 6);
        const bool hthread = tid < SG_LH * 4;
        const int hr = tid >> 2, hc0 = (tid & 3) * 8;
        const int c = tid & 31, r0 = (tid >> 5) * 4;
        const float bgc = bg[ch];
        float gq[3][4];
        if (flat) {                                         // (one chain per map: see sg_ssim_stats_kernel)
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const float v = q == 0 ? f0 : q == 1 ? f1 : f2;
                float h = 0.0f, t = 0.0f;
#pragma unroll
                for (int k = 0; k < 11; k++) h = fmaf(a.w[k], v, h);
#pragma unroll
                for (int k = 0; k < 11; k++) t = fmaf(a.w[k], h, t);
#pragma unroll
                for (int j = 0; j < 4; j++) gq[q][j] = t;
            }
        } else
#pragma unroll
        for (int q = 0; q < 3; q++) {
            if (hthread) {
                float v[18];
#pragma unroll
                for (int k = 0; k < 18; k++) v[k] = sM[q][hr][hc0 + k];
#pragma unroll
                for (int o = 0; o < 8; o++) {
                    float h = 0.0f;
#pragma unroll
                    for (int k = 0; k < 11; k++) h = fmaf(a.w[k], v[o + k], h);
                    sH[hr][hc0 + o] = h;
                }
            }
            __syncthreads();
            float col[14];
#pragma unroll
            for (int k = 0; k < 14; k++) col[k] = sH[r0 + k][c];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float h = 0.0f;
#pragma unroll
                for (int k = 0; k < 11; k++) h = fmaf(a.w[k], col[j + k], h);
                gq[q][j] = h;
            }
            __syncthreads();
        }
        float trv[4], tm[4], tgt[4];                        // the rendered / mask / target pixels of the four rows: twelve loads, one round trip
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int x = X0 + c, y = Y0 + r0 + j;
            const int xc = x < a.W ? x : a.W - 1, yc = y < a.H ? y : a.H - 1;
            const size_t p = (size_t)yc * a.W + xc;
            trv[j] = raw[ch * hw + p]; tm[j] = mask[p]; tgt[j] = gt_rgb[ch * hw + p];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int r = r0 + j;
            const int x = X0 + c, y = Y0 + r;
            const float g0 = gq[0][j], g1 = gq[1][j], g2 = gq[2][j];
            if (x < a.W && y < a.H) {
                const size_t p = (size_t)y * a.W + x;
                const float rv = trv[j], m = tm[j];
                const float xv = sg_clamp01(rv), yv = tgt[j] * m + bgc * (1.0f - m);
                const float d = xv - yv;
                const float sgn = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
                float g = c_ss * (g0 + 2.0f * xv * g1 + yv * g2) + c_l1 * sgn;
                // torch.clamp passes the gradient where min <= x <= max
                dL_draw[ch * hw + p] = (rv >= 0.0f && rv <= 1.0f) ? g : 0.0f;
            }
        }
    }
}

size_t sg_photo_loss_ws_bytes_impl(int W, int H)
{
    const size_t hw = (size_t)W * H;
    const size_t nb = (size_t)((W + SG_LT - 1) / SG_LT) * ((H + SG_LT - 1) / SG_LT) * 3;
    return sg_align(9 * hw * 4) + sg_align(nb * 16) + 256;
}

static SgLossArgs sg_loss_args(int W, int H, float l1_w, float ssim_w, size_t gt_stride = 0, size_t mask_stride = 0, int up_stride = 0)
{
    SgLossArgs a;
    a.W = W; a.H = H; a.l1_w = l1_w; a.ssim_w = ssim_w; a.up_stride = up_stride;
    a.gt_stride = gt_stride; a.mask_stride = mask_stride; a.ws_stride = sg_photo_loss_ws_bytes_impl(W, H);
    {   // losses/utils.py:28-30 in fp32, like torch.Tensor([...]) / sum()
        float g[11], s = 0.0f;
        for (int x = 0; x < 11; x++) { g[x] = (float)exp(-(double)((x - 5) * (x - 5)) / (2.0 * 1.5 * 1.5)); }
        for (int x = 0; x < 11; x++) s += g[x];
        for (int x = 0; x < 11; x++) a.w[x] = g[x] / s;
    }
    return a;
}

// gradient pass alone over the workspace of an earlier forward-only call (window statistics + scalars)
void sg_launch_photo_loss_bwd(int K, int W, int H, float l1_w, float ssim_w, const float *raw, const float *gt_rgb,
                              const float *mask, const float *bg, const void *ws, const float *upstream, float *dL_draw,
                              size_t gt_stride, size_t mask_stride, hipStream_t st, int up_stride)
{
    const SgLossArgs a = sg_loss_args(W, H, l1_w, ssim_w, gt_stride, mask_stride, up_stride);
    const size_t hw = (size_t)W * H;
    dim3 grid((W + SG_LT - 1) / SG_LT, (H + SG_LT - 1) / SG_LT, 3 * K), block(256);
    const int nb = (int)(grid.x * grid.y * 3);
    const char *b = (const char *)ws;
    const float *maps = (const float *)b;
    const float *scalars = (const float *)(b + sg_align(9 * hw * 4) + sg_align((size_t)nb * 16));
    sg_prof_begin(SG_K_PHOTO_LOSS, st);
    hipLaunchKernelGGL(sg_ssim_grad_kernel, grid, block, 0, st, a, raw, gt_rgb, mask, bg, maps, scalars, upstream, dL_draw);
    sg_prof_end(SG_K_PHOTO_LOSS, st);
}

// K frames per launch: three launches for the K losses and gradients (frame = blockIdx.z / 3), K = 1: the single-frame call
void sg_launch_photo_loss(int K, int W, int H, float l1_w, float ssim_w, const float *raw, const float *gt_rgb,
                          const float *mask, const float *bg, void *ws, float *pred_out, float *gt_out,
                          float *losses, const float *upstream, float *dL_draw, size_t gt_stride, size_t mask_stride, hipStream_t st)
{
    const SgLossArgs a = sg_loss_args(W, H, l1_w, ssim_w, gt_stride, mask_stride);
    const size_t hw = (size_t)W * H;
    dim3 grid((W + SG_LT - 1) / SG_LT, (H + SG_LT - 1) / SG_LT, 3 * K), block(256);
    const int nb = (int)(grid.x * grid.y * 3);
    char *b = (char *)ws;
    float *maps = (float *)b;
    float4 *partial = (float4 *)(b + sg_align(9 * hw * 4));
    float *scalars = (float *)(b + sg_align(9 * hw * 4) + sg_align((size_t)nb * 16));
    sg_prof_begin(SG_K_PHOTO_LOSS, st);
    hipLaunchKernelGGL(sg_ssim_stats_kernel, grid, block, 0, st, a, raw, gt_rgb, mask, bg, maps, pred_out, gt_out, partial);
    hipLaunchKernelGGL(sg_loss_reduce_kernel, dim3(K), dim3(256), 0, st, a, partial, nb, scalars, losses);
    if (dL_draw)
        hipLaunchKernelGGL(sg_ssim_grad_kernel, grid, block, 0, st, a, raw, gt_rgb, mask, bg, maps, scalars, upstream, dL_draw);
    sg_prof_end(SG_K_PHOTO_LOSS, st);
}
