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

struct SgLossArgs {
    int W, H;
    float l1_w, ssim_w;
    float w[11];                  // the reference's fp32 window: exp(-(x-5)^2 / (2 * 1.5^2)), normalised
    // K frames per launch (round 4): blockIdx.z = 3 frame + channel.  Frame f: raw / gradient / optional images at + 3 H W f,
    // target at + gt_stride f, mask at + mask_stride f (floats; 0 = one target / mask for all frames), workspace at + ws_stride f
    // bytes, losses at + 4 f
    size_t gt_stride, mask_stride, ws_stride;
};

__device__ __forceinline__ float sg_clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

// block partial layout: [block][4] = (sum |pred - gt|, sum ssim, sum mask, unused)
__global__ void __launch_bounds__(256)
sg_ssim_stats_kernel(SgLossArgs a, const float *__restrict__ raw, const float *__restrict__ gt_rgb,
                     const float *__restrict__ mask, const float *__restrict__ bg, float *__restrict__ maps,
                     float *__restrict__ pred_out, float *__restrict__ gt_out, float4 *__restrict__ partial)
{
    __shared__ float sX[SG_LH][SG_LP], sY[SG_LH][SG_LP];
    __shared__ float sH[SG_LH][SG_LT + 1];
    __shared__ float sRed[4][3];
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
        for (int i = tid; i < SG_LH * SG_LH; i += 256) {
            const int r = i / SG_LH, c = i - r * SG_LH;
            const int x = X0 - 5 + c, y = Y0 - 5 + r;
            float xv = 0.0f, yv = 0.0f;
            if (x >= 0 && x < a.W && y >= 0 && y < a.H) {
                const size_t p = (size_t)y * a.W + x;
                const float m = mask[p];
                xv = sg_clamp01(raw[ch * hw + p]);
                yv = gt_rgb[ch * hw + p] * m + bgc * (1.0f - m);
            }
            sX[r][c] = xv; sY[r][c] = yv;
        }
        __syncthreads();
        // Separable window, one quantity at a time through ONE LDS plane (20 KB per workgroup instead of 42 KB: the
        // kernel is latency-bound, resident workgroups are what hides it).  Horizontal pass: an item = 8 consecutive
        // output columns of one halo row, its 18 samples of x and y stay in registers for all five quantities;
        // vertical pass: 4 consecutive rows of one column per thread (14 LDS reads for 4 outputs).
        const bool hthread = tid < SG_LH * 4;
        const int hr = tid >> 2, hc0 = (tid & 3) * 8;
        float xs[18], ys[18];
        if (hthread) {
#pragma unroll
            for (int k = 0; k < 18; k++) { xs[k] = sX[hr][hc0 + k]; ys[k] = sY[hr][hc0 + k]; }
        }
        const int c = tid & 31, r0 = (tid >> 5) * 4;
        float vq[5][4];
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
                if (ch == 0) acc_mask += mask[p];
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
    const float u_l1 = upstream ? upstream[0] : 1.0f, u_ss = upstream ? upstream[1] : 1.0f;
    const float c_l1 = scalars[4] * u_l1, c_ss = scalars[5] * u_ss;
    {
        const int ch = blockIdx.z - 3 * frame;
        for (int i = tid; i < SG_LH * SG_LH; i += 256) {
            const int r = i / SG_LH, c = i - r * SG_LH;
            const int x = X0 - 5 + c, y = Y0 - 5 + r;
            float m0 = 0.0f, m1 = 0.0f, m2 = 0.0f;
            if (x >= 0 && x < a.W && y >= 0 && y < a.H) {
                const size_t p = (size_t)y * a.W + x;
                m0 = maps[(size_t)(ch * 3 + 0) * hw + p]; m1 = maps[(size_t)(ch * 3 + 1) * hw + p];
                m2 = maps[(size_t)(ch * 3 + 2) * hw + p];
            }
            sM[0][r][c] = m0; sM[1][r][c] = m1; sM[2][r][c] = m2;
        }
        __syncthreads();
        const bool hthread = tid < SG_LH * 4;
        const int hr = tid >> 2, hc0 = (tid & 3) * 8;
        const int c = tid & 31, r0 = (tid >> 5) * 4;
        const float bgc = bg[ch];
        float gq[3][4];
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
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int r = r0 + j;
            const int x = X0 + c, y = Y0 + r;
            const float g0 = gq[0][j], g1 = gq[1][j], g2 = gq[2][j];
            if (x < a.W && y < a.H) {
                const size_t p = (size_t)y * a.W + x;
                const float rv = raw[ch * hw + p], m = mask[p];
                const float xv = sg_clamp01(rv), yv = gt_rgb[ch * hw + p] * m + bgc * (1.0f - m);
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

static SgLossArgs sg_loss_args(int W, int H, float l1_w, float ssim_w, size_t gt_stride = 0, size_t mask_stride = 0)
{
    SgLossArgs a;
    a.W = W; a.H = H; a.l1_w = l1_w; a.ssim_w = ssim_w;
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
                              size_t gt_stride, size_t mask_stride, hipStream_t st)
{
    const SgLossArgs a = sg_loss_args(W, H, l1_w, ssim_w, gt_stride, mask_stride);
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
