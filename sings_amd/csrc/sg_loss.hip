// Photometric loss of one rendered view, forward + gradient in ONE pass over the image.
//
// Replaces, for the L1 and SSIM terms of HumanSceneLoss.forward (sings/rec/losses/loss.py:55-69), the chain
//   torch.clamp(rendered, 0, 1)                                  gs_renderer_single.py:96
//   gt = rgb * mask + bg * (1 - mask)                             loss.py:58
//   l1_w * |pred - gt|.sum() / mask.sum()                         losses/utils.py:16-20
//   ssim_w * (1 - ssim(pred, gt)) * mask.sum() / (H W)            losses/utils.py:28-70, loss.py:65-67
// and their autograd backward down to dL/d(rasterizer output).  (The LPIPS term is a VGG network: out of scope.)
//
// The reference runs ~25 elementwise / conv2d kernels each way.  Rounds 3-5 ran two tiled kernels with three fp32 derivative
// maps per channel between them (36 B/pixel written, 62 B/pixel read back with the halo: 121 us per 1080p view, 5 x its HBM
// bound, ten workgroup barriers per tile).  Round 6: the window statistics, their derivatives and the gradient window are ONE
// marching kernel and nothing but the image planes touches memory:
//   sg_mask_sum_kernel   : SG_NP fixed-order partial sums of the mask (the gradient scales need mask.sum() up front)
//   sg_photo_kernel      : one WAVE per (channel, strip of 64 columns, chunk of R rows), marching down the rows.  Per step it loads
//                          one row (coalesced, prefetched one step ahead), applies the 11-tap window ACROSS lanes through a
//                          wave-private LDS row, and DOWN the rows in registers: the last 11 row results of the five window
//                          quantities are a register ring (the row loop is unrolled by 11 so the ring index is static).  Five
//                          rows later the SSIM value and its three partial derivatives exist for that row; they go through a
//                          second register ring (vertical window) and one more LDS row (horizontal window) and leave as the
//                          gradient ten rows behind the loads.  No workgroup barrier anywhere, no float atomics.
//   sg_loss_reduce_kernel: fixed-order sum of the per-wave partials -> the four loss scalars
// A strip holds statistics for 64 columns and emits gradients for the 54 inner ones (the gradient window needs +-5 columns of
// derivatives); a chunk of R output rows reads R + 20 input rows.  Algorithmic bytes per pixel = 3 ch x (raw 4 + gt 4 + grad 4)
// + mask 4 = 40; the halo re-reads (x 1.19 columns, x (R + 20) / R rows) and the second read of the centre row are L2 / Infinity
// Cache hits.  Every pixel's sums run in the same order over its own 21 x 21 neighbourhood, so pixels with identical
// neighbourhoods get identical bits wherever they lie in a strip (tests: flat background of an avatar frame).
#include "sg_common.h"

#define SG_LN 64                  // lanes of a strip = columns whose window statistics it holds
#define SG_NP 256                 // mask partial sums per frame
#define SG_LOSS_UNITS 2048        // units (workgroups) a launch aims for: eight per CU, all resident
#define SG_LOSS_RUN 12            // consecutive units (3 channels x 4 strips) dealt to one XCD
// reciprocals of the two SSIM denominators: v_rcp_f32 (1 ulp) instead of the correctly rounded division (ten instructions each);
// the difference is far inside the fp32 noise of the window sums in front of it
#define SG_LOSS_RCP(v) __builtin_amdgcn_rcpf(v)

struct SgLossArgs {
    int W, H;
    float l1_w, ssim_w;
    float w[11];                  // the reference's fp32 window: exp(-(x-5)^2 / (2 * 1.5^2)), normalised
    // K frames per launch: blockIdx.y = frame.  Frame f: raw / gradient / optional images at + 3 H W f, target at + gt_stride f,
    // mask at + mask_stride f (floats; 0 = one target / mask for all frames), workspace at + ws_stride f bytes, losses at + 4 f
    size_t gt_stride, mask_stride, ws_stride;
    int up_stride;                // floats between the upstream weight pairs of consecutive frames: 0 (shared) | 2
    int R, nstrips, nchunks, units;   // rows per chunk, strips per row of chunks, chunks, waves per frame (3 nstrips nchunks)
    int xcd_rot;                  // frame f deals its runs of units to the XCDs starting at XCD f * xcd_rot (sg_loss_unit_of_block)
};

// workspace of one frame: [SG_NP floats: mask partials][cap float4: per-wave (sum |pred - gt|, sum ssim)][8 floats: scalars]
__host__ __device__ inline size_t sg_loss_unit_cap(int W) { return (size_t)SG_LOSS_UNITS + 3 * (size_t)((W + 53) / 54) + 8; }
#define SG_LOSS_OFF_PART ((size_t)SG_NP * 4)
__host__ __device__ inline size_t sg_loss_off_scalars(int W) { return SG_LOSS_OFF_PART + ((sg_loss_unit_cap(W) * 16 + 255) & ~(size_t)255); }

typedef float sg_f2 __attribute__((ext_vector_type(2)));
typedef float sg_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sg_clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

// LDS hand-off inside ONE wave: the LDS executes a wave's instructions in order, only the compiler must not move the loads up
__device__ __forceinline__ void sg_wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double sg_wave_sum_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// mask.sum() of one frame from its SG_NP partials, the same operations in every wave that needs it (bitwise the same value)
__device__ __forceinline__ double sg_mask_total(const float *__restrict__ mpart, int lane)
{
    const float4 p = ((const float4 *)mpart)[lane];
    return sg_wave_sum_f64(((double)p.x + (double)p.y) + ((double)p.z + (double)p.w));
}

// 1024 threads x 8 independent loads: the 8 MB of a 1080p mask are a latency problem (256 threads x 32 dependent trips took 7.2 us)
__global__ void __launch_bounds__(1024)
sg_mask_sum_kernel(SgLossArgs a, const float *__restrict__ mask, char *__restrict__ ws)
{
    __shared__ float sR[16];
    const size_t hw = (size_t)a.W * a.H;
    const int frame = blockIdx.y;
    mask += (size_t)frame * a.mask_stride;
    float *mpart = (float *)(ws + (size_t)frame * a.ws_stride);
    const size_t per = (hw + SG_NP - 1) / SG_NP, i0 = (size_t)blockIdx.x * per, i1 = i0 + per < hw ? i0 + per : hw;
    float acc = 0.0f;
    for (size_t i = i0 + threadIdx.x; i < i1; i += 8 * 1024) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { const size_t j = i + (size_t)k * 1024; v[k] = j < i1 ? mask[j] : 0.0f; }
        acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sR[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
        for (int k = 0; k < 16; k++) t += sR[k];
        mpart[blockIdx.x] = t;
    }
}

// image planes are addressed as raw buffers: descriptor in SGPRs, the row's byte offset in an SGPR, the lane's column offset in a
// VGPR that never changes -- no vector instruction goes into addressing
typedef __amdgpu_buffer_rsrc_t sg_rsrc;
__device__ __forceinline__ sg_rsrc sg_make_rsrc(const void *p, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)(uint32_t)bytes, 0x00020000);
}
__device__ __forceinline__ float sg_buf_ld(sg_rsrc r, uint32_t voff, uint32_t soff)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void sg_buf_st(sg_rsrc r, uint32_t voff, uint32_t soff, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ int sg_clampi(int v, int hi) { return v < 0 ? 0 : (v < hi ? v : hi - 1); }

// what a wave of a unit knows: its rows, its lane's columns, the planes of its channel
struct SgLossCtx {
    sg_rsrc raw, gt, mask;        // channel planes of this frame
    sg_rsrc grad, pred_out, gt_out;
    bool has_pred, has_gt;
    sg_f2 *sIn;                   // [2][SG_LN + 16]   staged rows (x, y) of the statistics wave
    sg_f4 *sD;                    // [2][SG_LN + 16]   derivative rows, statistics wave -> gradient wave (entry lane + 5)
    int H, R0, R1, lane;
    uint32_t pitch;               // bytes per image row
    uint32_t o0, o1, oc;          // byte offsets in a row: the two columns a lane loads for the row window, its own column
    bool in0, in1, incol, out_lane;   // those columns lie inside the image; this lane emits results (inner lane, column inside)
    float bgc, c_l1, c_ss;
};

template <bool GRAD> struct SgLossGeo {
    static constexpr int LAG = GRAD ? 10 : 5;     // rows between the loads and the centre (output) row
    static constexpr int OUT0 = GRAD ? 5 : 0;     // first lane that emits results
    static constexpr int NOUT = GRAD ? 54 : 64;   // lanes that emit results = strip pitch in columns
};

// register state of the statistics wave
struct SgStatSt {
    float hw[5][11];              // ring of the last 11 rows of the horizontally windowed x, y, x^2, y^2, x y
    float pin[2][6];              // the next two input rows, in flight: raw, target, mask at the lane's column, and at the one 64 further (lanes 0-9)
    float pc[3];                  // forward-only kernel: the next centre row in flight (raw, target, mask at the lane's own column)
    float acc_l1, acc_ssim;
};
// loads of row y (clamped into the image: rows outside count as zeros where they are USED), all lanes, no branches
__device__ __forceinline__ void sg_loss_prefetch_row(const SgLossCtx &c, float (&pin)[6], int y)
{
    const uint32_t r = (uint32_t)sg_clampi(y, c.H) * c.pitch;
    pin[0] = sg_buf_ld(c.raw, c.o0, r); pin[1] = sg_buf_ld(c.gt, c.o0, r); pin[2] = sg_buf_ld(c.mask, c.o0, r);
    pin[3] = sg_buf_ld(c.raw, c.o1, r); pin[4] = sg_buf_ld(c.gt, c.o1, r); pin[5] = sg_buf_ld(c.mask, c.o1, r);
}
__device__ __forceinline__ void sg_loss_prefetch_centre(const SgLossCtx &c, float (&pc)[3], int y)
{
    const uint32_t r = (uint32_t)sg_clampi(y, c.H) * c.pitch;
    pc[0] = sg_buf_ld(c.raw, c.oc, r); pc[1] = sg_buf_ld(c.gt, c.oc, r); pc[2] = sg_buf_ld(c.mask, c.oc, r);
}

// The window weights live in VECTOR registers.  A vector instruction with an SGPR source issues at half rate on gfx950 (tools/fmac_probe.hip:
// v_fmac v,v,v 2.6 cycles per wave-instruction per SIMD, v_fmac v,s,v 4.2 -- as do v_max / v_min / v_cmp / v_cndmask and packed fp32), and nine
// tenths of this file's instructions are w[k] * value.  The symmetric window has six distinct values; the asm keeps the compiler from
// recognising them as uniform and putting them back into SGPRs.
struct SgW { float w[11]; };
__device__ __forceinline__ SgW sg_loss_weights(const SgLossArgs &a)
{
    SgW w;
#pragma unroll
    for (int k = 0; k < 6; k++) { float v; asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(a.w[k])); w.w[k] = v; w.w[10 - k] = v; }
    return w;
}

// 11-tap sum in the fixed order k = 0..10 (the first tap is a product: fma(w, v, 0) without the zero)
#define SG_WIN11(acc, expr) { acc = w.w[0] * (expr(0)); acc = fmaf(w.w[1], expr(1), acc); acc = fmaf(w.w[2], expr(2), acc);      \
    acc = fmaf(w.w[3], expr(3), acc); acc = fmaf(w.w[4], expr(4), acc); acc = fmaf(w.w[5], expr(5), acc);                        \
    acc = fmaf(w.w[6], expr(6), acc); acc = fmaf(w.w[7], expr(7), acc); acc = fmaf(w.w[8], expr(8), acc);                        \
    acc = fmaf(w.w[9], expr(9), acc); acc = fmaf(w.w[10], expr(10), acc); }

// One row step of the statistics wave.  PH = t mod 11 is the ring slot the new row goes to; slot (PH + 1 + k) mod 11 holds the row
// k rows below the oldest.  GRAD: the derivative row of ys = yin - 5 goes to sD[t & 1] for the gradient wave (t >= 10);
// otherwise (forward-only kernel) this wave also sums |x - y| over the centre row and writes the optional images.
template <bool GRAD, bool LOSS, int PH>
__device__ __forceinline__ void sg_stat_step(const SgLossArgs &a, const SgW &w, const SgLossCtx &c, SgStatSt &s, int t)
{
    using G = SgLossGeo<GRAD>;
    const int yin = c.R0 - G::LAG + t;
    const int buf = t & 1;
    sg_f2 *sIn = c.sIn + buf * (SG_LN + 16);
    // (1) the row that arrived -> clamped render x, composited target y (zero outside the image: conv2d's zero padding)
    {
        const bool rowok = (unsigned)yin < (unsigned)c.H;
        const bool k0 = rowok && c.in0, k1 = rowok && c.in1;
        const float m0 = s.pin[0][2], m1 = s.pin[0][5];
        const float xv0 = k0 ? sg_clamp01(s.pin[0][0]) : 0.0f, yv0 = k0 ? s.pin[0][1] * m0 + c.bgc * (1.0f - m0) : 0.0f;
        const float xv1 = k1 ? sg_clamp01(s.pin[0][3]) : 0.0f, yv1 = k1 ? s.pin[0][4] * m1 + c.bgc * (1.0f - m1) : 0.0f;
        sIn[c.lane] = sg_f2{ xv0, yv0 };
        if (c.lane < 10) sIn[SG_LN + c.lane] = sg_f2{ xv1, yv1 };
    }
    // two rows stay in flight: the memory latency under load is longer than the half step between these loads and the next step's head
#pragma unroll
    for (int q = 0; q < 6; q++) s.pin[0][q] = s.pin[1][q];
    sg_loss_prefetch_row(c, s.pin[1], yin + 2);
    sg_wave_lds_sync();
    // (2) 11-tap window across the lanes
    {
        sg_f2 v[11];
#pragma unroll
        for (int k = 0; k < 11; k++) v[k] = sIn[c.lane + k];
        float h0, h1, h2, h3, h4;
        {
            const float wx = w.w[0] * v[0].x, wy = w.w[0] * v[0].y;
            h0 = wx; h1 = wy; h2 = wx * v[0].x; h3 = wy * v[0].y; h4 = wx * v[0].y;
        }
#pragma unroll
        for (int k = 1; k < 11; k++) {
            const float wx = w.w[k] * v[k].x, wy = w.w[k] * v[k].y;
            h0 += wx; h1 += wy;
            h2 = fmaf(wx, v[k].x, h2); h3 = fmaf(wy, v[k].y, h3); h4 = fmaf(wx, v[k].y, h4);
        }
        s.hw[0][PH] = h0; s.hw[1][PH] = h1; s.hw[2][PH] = h2; s.hw[3][PH] = h3; s.hw[4][PH] = h4;
    }
    if (t < 10) return;
    // (3) window down the rows: statistics of row ys, the SSIM value and its partial derivatives there
    const int ys = yin - 5;
    {
        float st[5];
#pragma unroll
        for (int q = 0; q < 5; q++) {
#define SG_RING(k) s.hw[q][(PH + 1 + (k)) % 11]
            SG_WIN11(st[q], SG_RING)
#undef SG_RING
        }
        const float mu1 = st[0], mu2 = st[1], e11 = st[2], e22 = st[3], e12 = st[4];
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        const float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
        const float s1 = e11 - mu1s, s2 = e22 - mu2s, s12 = e12 - mu12;
        const float A = mu1s + mu2s + C1, B = s1 + s2 + C2, Cn = 2.0f * mu12 + C1, D = 2.0f * s12 + C2;
        const float rA = SG_LOSS_RCP(A), rB = SG_LOSS_RCP(B);
        const float m = (Cn * D) * (rA * rB);
        const bool rowok = (unsigned)ys < (unsigned)c.H;
        if (GRAD) {
            // partials of m with respect to the window outputs mu1 = w*x, E[x^2] = w*x^2, E[xy] = w*xy; no pixel, no derivative
            const float dm_ds1 = -m * rB;
            const float dm_ds12 = 2.0f * Cn * (rA * rB);
            const float dm_dmu1 = 2.0f * mu2 * D * (rA * rB) - 2.0f * mu1 * m * rA - 2.0f * mu1 * dm_ds1 - mu2 * dm_ds12;
            const bool inimg = rowok && c.incol;
            c.sD[buf * (SG_LN + 16) + c.lane + 5] = sg_f4{ inimg ? dm_dmu1 : 0.0f, inimg ? dm_ds1 : 0.0f, inimg ? dm_ds12 : 0.0f, 0.0f };
        }
        if (LOSS) s.acc_ssim += (c.out_lane && rowok && ys >= c.R0 && ys < c.R1) ? m : 0.0f;
    }
    if (!GRAD) {
        // forward only: the centre row is the statistics row
        const float mk = s.pc[2];
        const float xv = sg_clamp01(s.pc[0]), yv = s.pc[1] * mk + c.bgc * (1.0f - mk);
        if (c.out_lane) {
            if (LOSS) s.acc_l1 += fabsf(xv - yv);
            const uint32_t r = (uint32_t)ys * c.pitch;
            if (c.has_pred) sg_buf_st(c.pred_out, c.oc, r, xv);
            if (c.has_gt) sg_buf_st(c.gt_out, c.oc, r, yv);
        }
        sg_loss_prefetch_centre(c, s.pc, ys + 1);
    }
}

// unit of workgroup b: runs of SG_LOSS_RUN consecutive units (the three channels of neighbouring strips: shared mask, shared halo
// columns) stay on one XCD's L2 (workgroup b runs on XCD b % 8), the runs are dealt round-robin to the XCDs.  A frame's runs are not a
// multiple of 8 in general (a 512 x 896 frame of a K = 8 launch: 20 runs -- XCDs 0-3 got three, XCDs 4-7 two, in EVERY frame: the
// launch took 3 / 2.5 of the balanced time); frame f therefore starts its deal at XCD f * (runs mod 8).
__device__ __forceinline__ int sg_loss_unit_of_block(int b, int shift)
{
    const int xcd = (b - shift) & 7, slot = b >> 3;
    return ((slot / SG_LOSS_RUN) * 8 + xcd) * SG_LOSS_RUN + slot % SG_LOSS_RUN;
}
static inline int sg_loss_blocks(int units) { return ((units + 8 * SG_LOSS_RUN - 1) / (8 * SG_LOSS_RUN)) * (8 * SG_LOSS_RUN); }

template <bool GRAD>
__device__ __forceinline__ void sg_loss_ctx(SgLossCtx &c, const SgLossArgs &a, int u, int frame, int lane, const float *raw, const float *gt_rgb,
                                            const float *mask, const float *bg, float *dL_draw, float *pred_out, float *gt_out)
{
    using G = SgLossGeo<GRAD>;
    const int ch = u % 3, strip = (u / 3) % a.nstrips, chunk = u / (3 * a.nstrips);
    const size_t hw = (size_t)a.W * a.H;
    c.H = a.H; c.lane = lane; c.pitch = (uint32_t)a.W * 4u;
    c.R0 = chunk * a.R; c.R1 = c.R0 + a.R < a.H ? c.R0 + a.R : a.H;
    {
        const int col = strip * G::NOUT - G::OUT0 + lane;        // the column whose window statistics this lane holds
        const int l0 = col - 5, l1 = l0 + SG_LN;                 // the columns it loads for the row window (the second: lanes 0-9)
        c.in0 = l0 >= 0 && l0 < a.W; c.in1 = l1 >= 0 && l1 < a.W; c.incol = col >= 0 && col < a.W;
        c.o0 = 4u * (uint32_t)sg_clampi(l0, a.W); c.o1 = lane < 10 ? 4u * (uint32_t)sg_clampi(l1, a.W) : c.o0;
        c.oc = 4u * (uint32_t)sg_clampi(col, a.W);
        c.out_lane = lane >= G::OUT0 && lane < G::OUT0 + G::NOUT && c.incol;
    }
    {
        const size_t plane = hw * 4, fo = (size_t)frame * 3 * hw + ch * hw;
        c.raw = sg_make_rsrc(raw + fo, plane); c.gt = sg_make_rsrc(gt_rgb + (size_t)frame * a.gt_stride + ch * hw, plane);
        c.mask = sg_make_rsrc(mask + (size_t)frame * a.mask_stride, plane);
        c.grad = sg_make_rsrc(GRAD ? dL_draw + fo : nullptr, GRAD ? plane : 0);
        c.has_pred = pred_out != nullptr; c.has_gt = gt_out != nullptr;
        c.pred_out = sg_make_rsrc(pred_out ? pred_out + fo : nullptr, pred_out ? plane : 0);
        c.gt_out = sg_make_rsrc(gt_out ? gt_out + fo : nullptr, gt_out ? plane : 0);
    }
    c.bgc = bg[ch];
    c.c_l1 = 0.0f; c.c_ss = 0.0f;
}

__device__ __forceinline__ void sg_stat_init(SgStatSt &s)
{
#pragma unroll
    for (int k = 0; k < 11; k++)
#pragma unroll
        for (int q = 0; q < 5; q++) s.hw[q][k] = 0.0f;
    s.pc[0] = s.pc[1] = s.pc[2] = 0.0f;
    s.acc_l1 = 0.0f; s.acc_ssim = 0.0f;
}

// forward only (the autograd forward): one wave per unit, 64 result columns per strip, chunks read R + 10 rows
template <bool LOSS>
__global__ void __launch_bounds__(64)
sg_photo_fwd_kernel(SgLossArgs a, const float *__restrict__ raw, const float *__restrict__ gt_rgb, const float *__restrict__ mask,
                    const float *__restrict__ bg, char *__restrict__ ws, float *__restrict__ pred_out, float *__restrict__ gt_out)
{
    __shared__ sg_f2 sIn[2][SG_LN + 16];
    const int u = sg_loss_unit_of_block(blockIdx.x, (int)blockIdx.y * a.xcd_rot);
    if (u >= a.units) return;
    const int lane = threadIdx.x, frame = blockIdx.y;
    ws += (size_t)frame * a.ws_stride;
    const SgW w = sg_loss_weights(a);
    SgLossCtx c;
    sg_loss_ctx<false>(c, a, u, frame, lane, raw, gt_rgb, mask, bg, nullptr, pred_out, gt_out);
    c.sIn = &sIn[0][0]; c.sD = nullptr;
    SgStatSt s;
    sg_stat_init(s);
    const int T = (c.R1 - c.R0) + 10;
    sg_loss_prefetch_row(c, s.pin[0], c.R0 - 5);
    sg_loss_prefetch_row(c, s.pin[1], c.R0 - 4);
    sg_loss_prefetch_centre(c, s.pc, c.R0);                  // (first used at t = 10, refreshed at the end of every later step)
    for (int t0 = 0; t0 < T; t0 += 11) {
#define SG_STEP(PH) if (t0 + PH < T) sg_stat_step<false, LOSS, PH>(a, w, c, s, t0 + PH)
        SG_STEP(0); SG_STEP(1); SG_STEP(2); SG_STEP(3); SG_STEP(4); SG_STEP(5); SG_STEP(6); SG_STEP(7); SG_STEP(8); SG_STEP(9); SG_STEP(10);
#undef SG_STEP
    }
    if (LOSS) {
        float a1 = s.acc_l1, a2 = s.acc_ssim;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a1 += __shfl_xor(a1, o, 64); a2 += __shfl_xor(a2, o, 64); }
        if (lane == 0) ((float4 *)(ws + SG_LOSS_OFF_PART))[u] = make_float4(a1, a2, 0.0f, 0.0f);
    }
}

// ---- forward + gradient: three waves per unit, a pipeline over the rows ------------------------------------------------------------
// As ONE wave the two rings cost 175 registers (two waves per SIMD) and every LDS round trip of a row sat on that wave's critical path:
// 105 us per 1080p view.  As a pipeline each wave runs a third of the instructions of a row and the kernel fits 96 registers:
//   wave H (rows 2i, 2i+1)    : loads, clamp / composite -> LDS rows -> 11-tap window across the lanes (x, y, x^2, y^2, x y) -> sH[i & 1]
//   wave V (rows of step i-1) : the 55-register ring, window down the rows, SSIM value and its three derivatives -> sD[(i - 1) & 1]
//   wave G (rows of step i-2) : adjoint window across the lanes straight from sD, the 33-register ring, window down the rows, gradient out
// One workgroup barrier per step hands the rows over (double-buffered: a slot is rewritten two barriers after it was read).  A step is TWO
// rows: a lone workgroup needs 1 200 cycles for a one-row step of ~100 vector instructions per wave (LDS round trips, the barrier, the
// loop) and 82 such steps in sequence bound the kernel from below; with two independent rows per step those latencies overlap.
#define SG_SIN_PITCH (SG_LN + 16)
// REPEATED ROWS.  Behind an avatar most of a frame is background on both sides (render = target = bg, bit for bit).  A row whose 74 staged
// (x, y) pairs are all equal, and equal to the row above, has the window sums of the row above: wave H hands those on and skips the LDS
// round trip and its 77 instructions per row.  The reused sums ARE what the skipped instructions would have produced (same operands, same
// order), so every pixel keeps the bits of the long way (tests: test_flat_tiles_take_the_short_path_with_the_same_bits).  (Carrying the
// flag on through waves V and G -- eleven repeated rows repeat the SSIM row -- was built too: it costs the kernel its 80 registers and a
// fast step is then as long as its row's memory round trip: LAB.md 6.1.)
struct SgHSt {
    float pin[2][6];              // the two rows of the NEXT step, in flight: raw, target, mask at the two columns of a lane
    uint32_t lasth[5];            // (uniform: SGPRs) window sums of the newest row, meaningful when that row held ONE (x, y) pair ...
    uint32_t lx, ly; bool lastflat;   // (uniform) ... these bits
};
struct SgVSt { float hw[5][11]; float acc_ssim; };
struct SgGSt { float dw[3][11]; float pc[2][3]; float acc_l1; };    // pc: the two centre rows of the next step, in flight

// clamp / composite of one prefetched row; returns (uniform) whether all 74 pairs of the row hold the bits (fx, fy)
__device__ __forceinline__ bool sg_h_stage(const SgLossCtx &c, const float (&pin)[6], int yin, float bgc, sg_f2 &own, sg_f2 &halo, uint32_t &fx, uint32_t &fy)
{
    const bool rowok = (unsigned)yin < (unsigned)c.H;
    const bool k0 = rowok && c.in0, k1 = rowok && c.in1;
    const float m0 = pin[2], m1 = pin[5];
    const float xv0 = k0 ? sg_clamp01(pin[0]) : 0.0f, yv0 = k0 ? pin[1] * m0 + bgc * (1.0f - m0) : 0.0f;
    const float xv1 = k1 ? sg_clamp01(pin[3]) : 0.0f, yv1 = k1 ? pin[4] * m1 + bgc * (1.0f - m1) : 0.0f;    // (lanes >= 10: their own column again)
    own = sg_f2{ xv0, yv0 }; halo = sg_f2{ xv1, yv1 };
    fx = __builtin_amdgcn_readfirstlane(__float_as_uint(xv0)); fy = __builtin_amdgcn_readfirstlane(__float_as_uint(yv0));
    return __all(__float_as_uint(xv0) == fx && __float_as_uint(yv0) == fy && __float_as_uint(xv1) == fx && __float_as_uint(yv1) == fy);
}

// step i of wave H: input rows R0 - 10 + 2i and the next one
__device__ __forceinline__ void sg_h_step(const SgLossArgs &a, const SgW &w, const SgLossCtx &c, SgHSt &s, sg_f2 *sIn, sg_f4 *sH4, float *sH1, int i)
{
    const int yin = c.R0 - 10 + 2 * i;
    const int buf = i & 1;
    sg_f2 *row0 = sIn + (buf * 2) * SG_SIN_PITCH, *row1 = row0 + SG_SIN_PITCH;
    sg_f2 own0, halo0, own1, halo1;
    uint32_t fx0, fy0, fx1, fy1;
    const bool flat0 = sg_h_stage(c, s.pin[0], yin, c.bgc, own0, halo0, fx0, fy0);
    const bool flat1 = sg_h_stage(c, s.pin[1], yin + 1, c.bgc, own1, halo1, fx1, fy1);
    sg_loss_prefetch_row(c, s.pin[0], yin + 2);              // (a whole step ahead: ~2 us under load)
    sg_loss_prefetch_row(c, s.pin[1], yin + 3);
    // a row repeats the one above: one pair in both, the same bits.  Both rows of the step repeat: nothing to do but hand the sums on
    const bool rep0 = flat0 && s.lastflat && fx0 == s.lx && fy0 == s.ly, rep1 = flat1 && flat0 && fx1 == fx0 && fy1 == fy0;
    s.lastflat = flat1; s.lx = fx1; s.ly = fy1;
    if (rep0 && rep1) {
#pragma unroll
        for (int r = 0; r < 2; r++) {
            sH4[(buf * 2 + r) * SG_LN + c.lane] = sg_f4{ __uint_as_float(s.lasth[0]), __uint_as_float(s.lasth[1]), __uint_as_float(s.lasth[2]),
                                                         __uint_as_float(s.lasth[3]) };
            sH1[(buf * 2 + r) * SG_LN + c.lane] = __uint_as_float(s.lasth[4]);
        }
        return;
    }
    row0[c.lane] = own0; row1[c.lane] = own1;
    if (c.lane < 10) { row0[SG_LN + c.lane] = halo0; row1[SG_LN + c.lane] = halo1; }
    sg_wave_lds_sync();
    // (x, y) pairs only: the three products are made per tap -- two more vector instructions per tap than reading them, but with five
    // values per entry the LDS was the busiest unit of the CU
    sg_f2 v0[11], v1[11];
#pragma unroll
    for (int k = 0; k < 11; k++) v0[k] = row0[c.lane + k];
#pragma unroll
    for (int k = 0; k < 11; k++) v1[k] = row1[c.lane + k];
    float h[2][5];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const sg_f2 *v = r ? v1 : v0;
        {
            const float wx = w.w[0] * v[0].x, wy = w.w[0] * v[0].y;
            h[r][0] = wx; h[r][1] = wy; h[r][2] = wx * v[0].x; h[r][3] = wy * v[0].y; h[r][4] = wx * v[0].y;
        }
#pragma unroll
        for (int k = 1; k < 11; k++) {
            const float wx = w.w[k] * v[k].x, wy = w.w[k] * v[k].y;
            h[r][0] += wx; h[r][1] += wy;
            h[r][2] = fmaf(wx, v[k].x, h[r][2]); h[r][3] = fmaf(wy, v[k].y, h[r][3]); h[r][4] = fmaf(wx, v[k].y, h[r][4]);
        }
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
        sH4[(buf * 2 + r) * SG_LN + c.lane] = sg_f4{ h[r][0], h[r][1], h[r][2], h[r][3] }; sH1[(buf * 2 + r) * SG_LN + c.lane] = h[r][4];
    }
#pragma unroll
    for (int q = 0; q < 5; q++) s.lasth[q] = __builtin_amdgcn_readfirstlane(__float_as_uint(h[1][q]));    // (one value across the lanes if the row was flat)
}

// statistics row ys = R0 - 15 + j from the ring whose newest entry (row j of wave H) sits in slot PH: SSIM value, its three derivatives -> sD row
template <bool LOSS, int PH>
__device__ __forceinline__ void sg_v_row(const SgW &w, const SgLossCtx &c, SgVSt &s, sg_f4 *drow, int j)
{
    const int ys = c.R0 - 15 + j;
    float st[5];
#pragma unroll
    for (int q = 0; q < 5; q++) {
#define SG_RING(k) s.hw[q][(PH + 1 + (k)) % 11]
        SG_WIN11(st[q], SG_RING)
#undef SG_RING
    }
    const float mu1 = st[0], mu2 = st[1], e11 = st[2], e22 = st[3], e12 = st[4];
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
    const float s1 = e11 - mu1s, s2 = e22 - mu2s, s12 = e12 - mu12;
    const float A = mu1s + mu2s + C1, B = s1 + s2 + C2, Cn = 2.0f * mu12 + C1, D = 2.0f * s12 + C2;
    const float rA = SG_LOSS_RCP(A), rB = SG_LOSS_RCP(B);
    const float m = (Cn * D) * (rA * rB);
    const bool rowok = (unsigned)ys < (unsigned)c.H;
    // partials of m with respect to the window outputs mu1 = w*x, E[x^2] = w*x^2, E[xy] = w*xy; no pixel, no derivative
    const float dm_ds1 = -m * rB;
    const float dm_ds12 = 2.0f * Cn * (rA * rB);
    const float dm_dmu1 = 2.0f * mu2 * D * (rA * rB) - 2.0f * mu1 * m * rA - 2.0f * mu1 * dm_ds1 - mu2 * dm_ds12;
    const bool inimg = rowok && c.incol;
    drow[c.lane + 5] = sg_f4{ inimg ? dm_dmu1 : 0.0f, inimg ? dm_ds1 : 0.0f, inimg ? dm_ds12 : 0.0f, 0.0f };
    if (LOSS) s.acc_ssim += (c.out_lane && rowok && ys >= c.R0 && ys < c.R1) ? m : 0.0f;
}

// step iv of wave V (P = iv mod 11): the two rows wave H left in sH[iv & 1] are rows j = 2 iv and 2 iv + 1 of the march
template <bool LOSS, int P>
__device__ __forceinline__ void sg_v_step(const SgLossArgs &a, const SgW &w, const SgLossCtx &c, SgVSt &s, const sg_f4 *sH4, const float *sH1, sg_f4 *sD, int iv)
{
    constexpr int PH0 = (2 * P) % 11, PH1 = (2 * P + 1) % 11;
    const int buf = iv & 1, j = 2 * iv;
    const sg_f4 ha = sH4[(buf * 2) * SG_LN + c.lane], hb = sH4[(buf * 2 + 1) * SG_LN + c.lane];
    const float ha4 = sH1[(buf * 2) * SG_LN + c.lane], hb4 = sH1[(buf * 2 + 1) * SG_LN + c.lane];
    s.hw[0][PH0] = ha.x; s.hw[1][PH0] = ha.y; s.hw[2][PH0] = ha.z; s.hw[3][PH0] = ha.w; s.hw[4][PH0] = ha4;
    if (j >= 10) sg_v_row<LOSS, PH0>(w, c, s, sD + (buf * 2) * SG_SIN_PITCH, j);
    s.hw[0][PH1] = hb.x; s.hw[1][PH1] = hb.y; s.hw[2][PH1] = hb.z; s.hw[3][PH1] = hb.w; s.hw[4][PH1] = hb4;
    if (j >= 10) sg_v_row<LOSS, PH1>(w, c, s, sD + (buf * 2 + 1) * SG_SIN_PITCH, j + 1);
}

// one derivative row of wave V (row j of the march) through the adjoint window: across the lanes straight from sD, into ring slot PH,
// and from j = 20 on down the rows -> gradient row yo = R0 - 20 + j
template <bool LOSS, int PH>
__device__ __forceinline__ void sg_g_row(const SgW &w, const SgLossCtx &c, SgGSt &s, const sg_f4 *drow, const float (&pc)[3], int j)
{
    {
        // (all 16 bytes of an entry are asked for: ds_read_b128 runs at twice the rate of the b96 the compiler would narrow this to)
#define SG_LDU(k) { u[k] = drow[c.lane + (k)]; asm volatile("" :: "v"(u[k].w)); }
        sg_f4 u[11];
        float g0, g1, g2;
        SG_LDU(0) SG_LDU(1) SG_LDU(2) SG_LDU(3) SG_LDU(4) SG_LDU(5) SG_LDU(6) SG_LDU(7) SG_LDU(8) SG_LDU(9) SG_LDU(10)
#undef SG_LDU
        g0 = w.w[0] * u[0].x; g1 = w.w[0] * u[0].y; g2 = w.w[0] * u[0].z;
#pragma unroll
        for (int k = 1; k < 11; k++) { g0 = fmaf(w.w[k], u[k].x, g0); g1 = fmaf(w.w[k], u[k].y, g1); g2 = fmaf(w.w[k], u[k].z, g2); }
        s.dw[0][PH] = g0; s.dw[1][PH] = g1; s.dw[2][PH] = g2;
    }
    if (j < 20) return;
    const int yo = c.R0 - 20 + j;
    float g[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
#define SG_RING(k) s.dw[q][(PH + 1 + (k)) % 11]
        SG_WIN11(g[q], SG_RING)
#undef SG_RING
    }
    const float rv = pc[0], mk = pc[2];
    const float xv = sg_clamp01(rv), yv = pc[1] * mk + c.bgc * (1.0f - mk);
    const float d = xv - yv;
    const bool out = c.out_lane && yo < c.R1;                // (an odd number of rows: the second row of the last step belongs to the next chunk)
    if (out) {
        const float sgn = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
        const float gr = c.c_ss * (g[0] + 2.0f * xv * g[1] + yv * g[2]) + c.c_l1 * sgn;
        const uint32_t r = (uint32_t)yo * c.pitch;
        sg_buf_st(c.grad, c.oc, r, (rv >= 0.0f && rv <= 1.0f) ? gr : 0.0f);        // torch.clamp passes the gradient where min <= x <= max
        if (c.has_pred) sg_buf_st(c.pred_out, c.oc, r, xv);
        if (c.has_gt) sg_buf_st(c.gt_out, c.oc, r, yv);
    }
    if (LOSS) s.acc_l1 += out ? fabsf(d) : 0.0f;
}

// step ig of wave G (P = ig mod 11): the two derivative rows in sD[ig & 1] are rows j = 2 ig and 2 ig + 1 of the march (called from j = 10 on)
template <bool LOSS, int P>
__device__ __forceinline__ void sg_g_step(const SgLossArgs &a, const SgW &w, const SgLossCtx &c, SgGSt &s, const sg_f4 *sD, int ig)
{
    constexpr int PH0 = (2 * P) % 11, PH1 = (2 * P + 1) % 11;
    const int buf = ig & 1, j = 2 * ig;
    sg_g_row<LOSS, PH0>(w, c, s, sD + (buf * 2) * SG_SIN_PITCH, s.pc[0], j);
    sg_g_row<LOSS, PH1>(w, c, s, sD + (buf * 2 + 1) * SG_SIN_PITCH, s.pc[1], j + 1);
    if (j >= 20) {
        // the centre rows (raw, target, mask at the lane's own column) of the next step
        const int yo = c.R0 - 20 + j;
        sg_loss_prefetch_centre(c, s.pc[0], yo + 2);
        sg_loss_prefetch_centre(c, s.pc[1], yo + 3);
    }
}

#ifdef SG_LOSS_STAMP
// diagnostic build only (tools/loss_stamps.py): per-wave start / end stamps of the shader clock and of the 100 MHz real-time clock
__device__ unsigned long long sg_loss_stamps[8192 * 3 * 4];
extern "C" int sg_debug_loss_stamps(unsigned long long *host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sg_loss_stamps), (size_t)n * 8); }
#define SG_STAMP_BEGIN const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#define SG_STAMP_END if (lane == 0 && blockIdx.y == 0) { unsigned long long *q = sg_loss_stamps + ((size_t)blockIdx.x * 3 + wave) * 4;      \
    q[0] = st_c0; q[1] = st_r0 | ((unsigned long long)(__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)) & 0xffffu) << 48) | ((unsigned long long)(__builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 0xfu) << 44); q[2] = __builtin_amdgcn_s_memtime(); q[3] = __builtin_amdgcn_s_memrealtime(); }
#else
#define SG_STAMP_BEGIN
#define SG_STAMP_END
#endif
// Time-sliced wave priority.  The arbiter serves the OLDEST wave first: of the eight workgroups of a CU (all start within 2 us) the first
// dispatched finished at 55 us and the last at 78, the CU half empty in between (tools/loss_stamps.py).  A priority that rotates every step
// gives every workgroup the same share: lifetimes 56-70 us, the launch 4.5 % shorter in cycles.
__device__ __forceinline__ void sg_rot_prio(int x)
{
    switch (x & 3) { case 0: __builtin_amdgcn_s_setprio(0); break; case 1: __builtin_amdgcn_s_setprio(1); break;
                     case 2: __builtin_amdgcn_s_setprio(2); break; default: __builtin_amdgcn_s_setprio(3); break; }
}
#define SG_PRIO(i) sg_rot_prio((i) + (int)blockIdx.x)
#define SG_BAR() __syncthreads()

template <bool LOSS>
__global__ void __launch_bounds__(192)
sg_photo_kernel(SgLossArgs a, const float *__restrict__ raw, const float *__restrict__ gt_rgb, const float *__restrict__ mask,
                const float *__restrict__ bg, char *__restrict__ ws, const float *__restrict__ upstream,
                float *__restrict__ dL_draw, float *__restrict__ pred_out, float *__restrict__ gt_out)
{
    __shared__ sg_f2 sIn[2 * 2][SG_SIN_PITCH];               // [step parity x row of the step]
    __shared__ sg_f4 sH4[2 * 2][SG_LN];
    __shared__ float sH1[2 * 2][SG_LN];
    __shared__ sg_f4 sD[2 * 2][SG_SIN_PITCH];
    SG_STAMP_BEGIN
    const int u = sg_loss_unit_of_block(blockIdx.x, (int)blockIdx.y * a.xcd_rot);
    if (u >= a.units) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), frame = blockIdx.y;
    ws += (size_t)frame * a.ws_stride;
    const SgW w = sg_loss_weights(a);
    SgLossCtx c;
    sg_loss_ctx<true>(c, a, u, frame, lane, raw, gt_rgb, mask, bg, dL_draw, pred_out, gt_out);
    c.sIn = nullptr; c.sD = nullptr;
    // rows wave H stages: R + 20, in steps of two (an odd last row is computed and dropped); V runs one step behind H, G two
    const int S = ((c.R1 - c.R0) + 20 + 1) / 2, I = S + 2;
    float part = 0.0f;
    if (wave == 0) {
        SgHSt s;
        s.lastflat = false; s.lx = s.ly = 0u;
#pragma unroll
        for (int q = 0; q < 5; q++) s.lasth[q] = 0u;
        sg_loss_prefetch_row(c, s.pin[0], c.R0 - 10); sg_loss_prefetch_row(c, s.pin[1], c.R0 - 9);
        for (int i = 0; i < I; i++) {
            SG_PRIO(i);
            if (i < S) sg_h_step(a, w, c, s, &sIn[0][0], &sH4[0][0], &sH1[0][0], i);
            SG_BAR();
        }
    } else if (wave == 1) {
        SgVSt s;
#pragma unroll
        for (int k = 0; k < 11; k++)
#pragma unroll
            for (int q = 0; q < 5; q++) s.hw[q][k] = 0.0f;
        s.acc_ssim = 0.0f;
        for (int i0 = 0; i0 < I; i0 += 11) {
            // step i handles the rows of wave H's step i - 1
#define SG_STEP(P) if (i0 + P < I) { SG_PRIO(i0 + P); if (i0 + P >= 1 && i0 + P <= S) sg_v_step<LOSS, (P + 10) % 11>(a, w, c, s, &sH4[0][0], &sH1[0][0], &sD[0][0], i0 + P - 1); SG_BAR(); }
            SG_STEP(0); SG_STEP(1); SG_STEP(2); SG_STEP(3); SG_STEP(4); SG_STEP(5); SG_STEP(6); SG_STEP(7); SG_STEP(8); SG_STEP(9); SG_STEP(10);
#undef SG_STEP
        }
        part = s.acc_ssim;
    } else {
        // gradient scales: d(l1_w |.|.sum() / mask.sum()) and d(ssim_w (1 - mean ssim) mask.sum() / (H W)), times the upstream weights
        // (d loss / d weighted l1 term, d loss / d weighted ssim term); 1, 1 when NULL
        {
            const size_t hw = (size_t)a.W * a.H;
            const double s_mask = sg_mask_total((const float *)ws, lane);
            const float *up = upstream ? upstream + (size_t)frame * a.up_stride : nullptr;
            const float u_l1 = up ? up[0] : 1.0f, u_ss = up ? up[1] : 1.0f;
            c.c_l1 = (float)((double)a.l1_w / s_mask) * u_l1;
            c.c_ss = (float)(-(double)a.ssim_w * (s_mask / (double)hw) / (3.0 * (double)hw)) * u_ss;
        }
        SgGSt s;
#pragma unroll
        for (int k = 0; k < 11; k++)
#pragma unroll
            for (int q = 0; q < 3; q++) s.dw[q][k] = 0.0f;
        s.acc_l1 = 0.0f;
        sg_loss_prefetch_centre(c, s.pc[0], c.R0); sg_loss_prefetch_centre(c, s.pc[1], c.R0 + 1);
        for (int i0 = 0; i0 < I; i0 += 11) {
            // step i handles the derivative rows of wave V's step i - 1 = wave H's step i - 2 (derivative rows exist from march row 10 on)
#define SG_STEP(P) if (i0 + P < I) { SG_PRIO(i0 + P); if (i0 + P >= 7) sg_g_step<LOSS, (P + 9) % 11>(a, w, c, s, &sD[0][0], i0 + P - 2); SG_BAR(); }
            SG_STEP(0); SG_STEP(1); SG_STEP(2); SG_STEP(3); SG_STEP(4); SG_STEP(5); SG_STEP(6); SG_STEP(7); SG_STEP(8); SG_STEP(9); SG_STEP(10);
#undef SG_STEP
        }
        part = s.acc_l1;
    }
    if (LOSS && wave != 0) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        // float4 (sum |pred - gt|, sum ssim, -, -) of the unit: each wave writes its own component
        if (lane == 0) ((float *)(ws + SG_LOSS_OFF_PART))[4 * u + (wave == 1 ? 1 : 0)] = part;
    }
    SG_STAMP_END
}

// scalars[0..3] = (l1_w * Ll1, ssim_w * Lssim, Ll1, mean ssim); scalars[4..5] = gradient scales (c_l1, c_ssim), for inspection
__global__ void __launch_bounds__(1024)
sg_loss_reduce_kernel(SgLossArgs a, char *__restrict__ ws, float *__restrict__ losses)
{
    __shared__ double sR[16][2];
    __shared__ double sMask;
    ws += (size_t)blockIdx.x * a.ws_stride;                   // frame blockIdx.x
    const float4 *partial = (const float4 *)(ws + SG_LOSS_OFF_PART);
    float *scalars = (float *)(ws + sg_loss_off_scalars(a.W));
    if (losses) losses += 4 * blockIdx.x;
    double t0 = 0, t1 = 0;
    for (int i = threadIdx.x; i < a.units; i += 1024) { const float4 p = partial[i]; t0 += p.x; t1 += p.y; }
    t0 = sg_wave_sum_f64(t0); t1 = sg_wave_sum_f64(t1);
    if ((threadIdx.x & 63) == 0) { sR[threadIdx.x >> 6][0] = t0; sR[threadIdx.x >> 6][1] = t1; }
    if (threadIdx.x < 64) { const double m = sg_mask_total((const float *)ws, threadIdx.x); if (threadIdx.x == 0) sMask = m; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s_l1 = 0, s_ssim = 0;
        for (int k = 0; k < 16; k++) { s_l1 += sR[k][0]; s_ssim += sR[k][1]; }
        const double hw = (double)a.W * a.H;
        const double s_mask = sMask;
        const double Ll1 = s_l1 / s_mask, ssim_mean = s_ssim / (3.0 * hw);
        const double Lssim = (1.0 - ssim_mean) * (s_mask / hw);
        scalars[0] = (float)(a.l1_w * Ll1); scalars[1] = (float)(a.ssim_w * Lssim);
        scalars[2] = (float)Ll1; scalars[3] = (float)ssim_mean;
        scalars[4] = (float)((double)a.l1_w / s_mask);
        scalars[5] = (float)(-(double)a.ssim_w * (s_mask / hw) / (3.0 * hw));
        if (losses) { losses[0] = scalars[0]; losses[1] = scalars[1]; losses[2] = scalars[2]; losses[3] = scalars[3]; }
    }
}

size_t sg_photo_loss_ws_bytes_impl(int W, int H)
{
    (void)H;
    return sg_loss_off_scalars(W) + 256;
}

static SgLossArgs sg_loss_args(int K, bool grad, int W, int H, float l1_w, float ssim_w, size_t gt_stride = 0, size_t mask_stride = 0,
                               int up_stride = 0)
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
    {   // strips x chunks: at most SG_LOSS_UNITS units per launch (one round of workgroups), chunks of at least 16 rows
        const int nout = grad ? SgLossGeo<true>::NOUT : SgLossGeo<false>::NOUT;
        a.nstrips = (W + nout - 1) / nout;
        const int cols = a.nstrips * 3 * K, most = (H + 15) / 16;
        int nch = SG_LOSS_UNITS / cols;
        nch = nch < 1 ? 1 : (nch > most ? most : nch);
        // (fewer, longer chunks that deal evenly to the XCDs -- 16 instead of 18 at 1080p, 40-48 instead of 56 at 512 x 896 -- measured:
        //  89.2 against 87.8 us, 37.7 against 39.5: the halo rows saved and the balance gained go to the emptier compute units)
        a.R = (H + nch - 1) / nch;
        a.nchunks = (H + a.R - 1) / a.R;
        a.units = a.nstrips * a.nchunks * 3;
        a.xcd_rot = ((a.units + SG_LOSS_RUN - 1) / SG_LOSS_RUN) % 8;
    }
    return a;
}

// gradient pass alone, after an earlier forward-only call on the same inputs (which left the mask partials in the workspace)
void sg_launch_photo_loss_bwd(int K, int W, int H, float l1_w, float ssim_w, const float *raw, const float *gt_rgb,
                              const float *mask, const float *bg, const void *ws, const float *upstream, float *dL_draw,
                              size_t gt_stride, size_t mask_stride, hipStream_t st, int up_stride)
{
    const SgLossArgs a = sg_loss_args(K, true, W, H, l1_w, ssim_w, gt_stride, mask_stride, up_stride);
    sg_prof_begin(SG_K_PHOTO_LOSS, st);
    hipLaunchKernelGGL((sg_photo_kernel<false>), dim3(sg_loss_blocks(a.units), K), dim3(192), 0, st, a, raw, gt_rgb, mask, bg,
                       (char *)ws, upstream, dL_draw, (float *)nullptr, (float *)nullptr);
    sg_prof_end(SG_K_PHOTO_LOSS, st);
}

// K frames per launch: three launches for the K losses and gradients (frame = blockIdx.y), K = 1: the single-frame call
void sg_launch_photo_loss(int K, int W, int H, float l1_w, float ssim_w, const float *raw, const float *gt_rgb,
                          const float *mask, const float *bg, void *ws, float *pred_out, float *gt_out,
                          float *losses, const float *upstream, float *dL_draw, size_t gt_stride, size_t mask_stride, hipStream_t st)
{
    const SgLossArgs a = sg_loss_args(K, dL_draw != nullptr, W, H, l1_w, ssim_w, gt_stride, mask_stride);
    const dim3 grid(sg_loss_blocks(a.units), K);
    sg_prof_begin(SG_K_PHOTO_LOSS, st);
    hipLaunchKernelGGL(sg_mask_sum_kernel, dim3(SG_NP, K), dim3(1024), 0, st, a, mask, (char *)ws);
    if (dL_draw)
        hipLaunchKernelGGL((sg_photo_kernel<true>), grid, dim3(192), 0, st, a, raw, gt_rgb, mask, bg, (char *)ws, upstream, dL_draw,
                           pred_out, gt_out);
    else
        hipLaunchKernelGGL((sg_photo_fwd_kernel<true>), grid, dim3(64), 0, st, a, raw, gt_rgb, mask, bg, (char *)ws, pred_out, gt_out);
    hipLaunchKernelGGL(sg_loss_reduce_kernel, dim3(K), dim3(1024), 0, st, a, (char *)ws, losses);
    sg_prof_end(SG_K_PHOTO_LOSS, st);
}
