// The decoders' linear layers on the matrix cores, bias and activation fused in:
//   forward   h = act(x W^T + b)                         modules/decoders.py:41-49, 75-94 (nn.Linear + GELU / Sigmoid)
//   backward  dz = dh * act'(z),  dx = dz W              autograd of the same lines
// replacing, per layer, a library GEMM + an element-wise pass each way (round 1: torch.addmm / torch.mm + sg_bias_act_*:
// 0.62 + 0.41 ms of the 2.96-ms training step).  fp32 in, fp32 out, v_mfma_f32_32x32x2_f32 (exact fp32 products).
//
// One kernel, two instantiations of the same tile loop:   Y[N, CO] = X'[N, CK] . M^T,   M [CO, CK]
//   forward : X' = x,                M = W   [Cout, Cin]      epilogue: z = Y + b, h = act(z)
//   backward: X' = dh * act'(z),     M = W^T [Cin, Cout]      prologue also stores dz (the weight-gradient kernel reads it)
// The forward leaves behind what makes act' ONE multiplication in the backward ("aux"): GELU -> gelu'(z) itself (the erf is
// evaluated once, in the forward's epilogue), sigmoid -> nothing (s' = h (1 - h) from the saved output), softplus -> z.
// Layer widths are <= 128, so the whole of M fits the register file of a workgroup: wave w keeps ITS 32 output columns
// of M -- CK/2 registers per lane, already in MFMA A-operand layout -- for the whole kernel; only the points stream.
// A tile of 32 PT points x CK inputs goes through LDS once (coalesced 16-B loads, shared by the four waves) and is read back
// as the B operand with one ds_read_b128 per four MFMAs: the reduction index is PERMUTED (MFMA step s multiplies inputs s
// and CK/2 + s), so the four values a lane needs next are adjacent.  The 32 x 32 result tiles leave straight from the
// accumulator registers: a lane holds four consecutive output columns of one point four times -> 16-B stores.  Loads of the
// next tile are in flight during the MFMAs; the waves of a workgroup meet only at the X' tile.
#include "sg_common.h"
#include <type_traits>

typedef float sg_v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float sgl_gelu(float z) { return 0.5f * z * (1.0f + erff(z * 0.70710678118654752440f)); }
__device__ __forceinline__ float sgl_gelu_grad(float z)
{
    const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752440f));
    return cdf + z * 0.39894228040143267794f * __expf(-0.5f * z * z);
}
// GELU and its derivative from ONE evaluation of erf: h = z Phi(z), h' = Phi(z) + z phi(z), Phi = (1 + erf(z / sqrt 2)) / 2.
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32 round-off of Phi): 1 - (a1 t + ... + a5 t^5) e^(-x^2),
// t = 1 / (1 + p |x|); the e^(-x^2) = e^(-z^2 / 2) it needs is the one phi(z) needs.
__device__ __forceinline__ void sgl_gelu_both(float z, float &h, float &dh)
{
    const float x = fabsf(z) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
    const float e = __expf(-x * x);
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float erfa = fmaf(-poly, e, 1.0f);                               // erf(|x|)
    const float cdf = 0.5f * (1.0f + copysignf(erfa, z));
    h = z * cdf;
    dh = fmaf(z * 0.39894228040143267794f, e, cdf);
}
// act: 0 identity, 1 GELU (erf), 2 sigmoid(z + row_offset), 3 log(exp(z) + 1)   (same codes as sg_bias_act_*)
__device__ __forceinline__ float sgl_act(int act, float z, float ro)
{
    if (act == 1) return sgl_gelu(z);
    if (act == 2) return 1.0f / (1.0f + expf(-(z + ro)));
    if (act == 3) return logf(expf(z) + 1.0f);
    return z;
}
// derivative from the forward's aux value: act 1: aux = gelu'(z); act 2: aux = h = sigmoid(.); act 3: aux = z
__device__ __forceinline__ float sgl_act_grad_aux(int act, float aux)
{
    if (act == 1) return aux;
    if (act == 2) return aux * (1.0f - aux);
    if (act == 3) { const float e = expf(aux); return e / (e + 1.0f); }
    return 1.0f;
}

#define SGL_MAXC 128
#define SGL_TILE 4352                // floats per LDS tile: 32 x (128 + 4), 64 x (64 + 4), 128 x (32 + 1) all fit (3 x 17 KiB per workgroup)
// 32-point tiles a workgroup handles per round: one wave per (column block, point tile), limited by the LDS tile
__host__ __device__ inline int sgl_point_tiles(int TO, int CKP)
{
    int pt = 4 / TO;
    while (pt > 1 && 32 * pt * (CKP + 4) > SGL_TILE) pt >>= 1;
    return pt;
}

// NQ = CKP / 8 (CKP = CK rounded up to a multiple of 8): float4 reads per lane and tile; BWD: the backward instantiation
template <int NQ, bool BWD>
__global__ void __launch_bounds__(256, 3)          // <= 168 VGPRs: three waves per SIMD (the 64 weight registers are the bulk)
sg_linear_kernel(int N, int CK, int CO, int act, const float *__restrict__ X, const float *__restrict__ Z,
                 const float *__restrict__ Mw, int m_co_stride, int m_ck_stride, const float *__restrict__ bias,
                 const float *__restrict__ row_offset, float *__restrict__ out0, float *__restrict__ out1)
{
    constexpr int CKP = NQ * 8, HALF = CKP / 2;
    __shared__ float sX[2][SGL_TILE];                                      // X' tile [32 PT points][CKP + 4], double-buffered; the Y tile
                                                                           // [32 PT points][32 TO + 1] reuses the buffer just consumed
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int TO = (CO + 31) / 32;                                         // 32-column blocks of the output (1..4)
    const int PT = sgl_point_tiles(TO, CKP);                               // 32-point tiles per round (waves beyond TO PT idle in the MFMA part)
    const int cb = wave % TO, pt = wave / TO;                              // this wave: column block, point tile
    const bool active = pt < PT;
    const int R = 32 * PT;                                                 // points per round
    const int XS = CKP + 4;
    const int j = lane & 31, h = lane >> 5;
    // ---- this wave's 32 columns of M as MFMA A operands: a[s] = M(32 cb + j, h HALF + s)
    float a[HALF];
    {
        const int co = 32 * cb + j;
        if (m_ck_stride == 1 && (m_co_stride & 3) == 0 && (CK & 3) == 0) {   // forward, row-major W: a lane's registers are consecutive in memory
#pragma unroll
            for (int s = 0; s < HALF; s += 4) {
                const int k = h * HALF + s;
                const float4 w4 = (active && co < CO && k < CK) ? *(const float4 *)(Mw + (size_t)co * m_co_stride + k)
                                                                 : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                a[s] = w4.x; a[s + 1] = w4.y; a[s + 2] = w4.z; a[s + 3] = w4.w;
            }
        } else {
#pragma unroll
            for (int s = 0; s < HALF; s++) {
                const int k = h * HALF + s;
                a[s] = (active && co < CO && k < CK) ? Mw[(size_t)co * m_co_stride + (size_t)k * m_ck_stride] : 0.0f;
            }
        }
    }
    // the 16 bias values of this lane's output columns (forward)
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int c = 32 * cb + 8 * (r >> 2) + 4 * h + (r & 3);
        bv[r] = (!BWD && bias && active && c < CO) ? bias[c] : 0.0f;
    }
    const int f4_per_row = CKP / 4;
    const int nf4 = R * f4_per_row;                                        // float4 elements of one X' tile
    constexpr int PTMAX = SGL_TILE / (32 * (CKP + 4)) >= 4 ? 4 : (SGL_TILE / (32 * (CKP + 4)) >= 2 ? 2 : 1);
    constexpr int PER = (PTMAX * 32 * CKP / 4 + 255) / 256;                // float4 per thread and tile (<= 4)
    const bool vec = (CK & 3) == 0;
    float4 xr[PER], zr[BWD ? PER : 1];
    // fetch: the loads only (they stay in flight during the MFMAs of the current tile); stash: act' (backward), dz out, LDS
    auto fetch = [&](int n0) {
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int f = tid + 256 * q;
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f), zz = v;
            if (f < nf4) {
                const int row = f / f4_per_row, c = 4 * (f - row * f4_per_row), n = n0 + row;
                if (n < N && c < CK) {
                    if (vec) {
                        v = *(const float4 *)(X + (size_t)n * CK + c);
                        if (BWD && act != 0) zz = *(const float4 *)(Z + (size_t)n * CK + c);
                    } else {
                        float e[4] = { 0.0f, 0.0f, 0.0f, 0.0f }, g[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
                        for (int u = 0; u < 4; u++)
                            if (c + u < CK) {
                                e[u] = X[(size_t)n * CK + c + u];
                                if (BWD && act != 0) g[u] = Z[(size_t)n * CK + c + u];
                            }
                        v = make_float4(e[0], e[1], e[2], e[3]); zz = make_float4(g[0], g[1], g[2], g[3]);
                    }
                }
            }
            xr[q] = v;
            if (BWD) zr[q] = zz;
        }
    };
    auto stash = [&](int buf, int n0) {
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int f = tid + 256 * q;
            if (f < nf4) {
                const int row = f / f4_per_row, c = 4 * (f - row * f4_per_row), n = n0 + row;
                float4 v = xr[q];
                if (BWD && act != 0) {
                    const float4 zz = zr[q];
                    v.x *= sgl_act_grad_aux(act, zz.x); v.y *= sgl_act_grad_aux(act, zz.y);
                    v.z *= sgl_act_grad_aux(act, zz.z); v.w *= sgl_act_grad_aux(act, zz.w);
                    if (out1 && n < N && c < CK) {                                                   // dz
                        if (vec) *(float4 *)(out1 + (size_t)n * CK + c) = v;
                        else {
                            const float e[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
                            for (int u = 0; u < 4; u++) if (c + u < CK) out1[(size_t)n * CK + c + u] = e[u];
                        }
                    }
                }
                *(float4 *)&sX[buf][row * XS + c] = v;
            }
        }
    };
    const int ntiles = (N + R - 1) / R;
    int tile = blockIdx.x;
    if (tile < ntiles) { fetch(tile * R); stash(0, tile * R); }
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(0x0F70);                                    // vmcnt(0): the weights have landed (see the wide kernel)
    int buf = 0;
    for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        const int n0 = tile * R, next = tile + gridDim.x;
        if (next < ntiles) fetch(next * R);                                // in flight during the MFMAs below
        sg_v16f acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.0f;
        if (active) {
            const float *xrow = &sX[buf][(32 * pt + j) * XS + h * HALF];   // B operand: X'[point j][h HALF + s]
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const float4 b = *(const float4 *)(xrow + 4 * q);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q], b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 1], b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 2], b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 3], b.w, acc, 0, 0, 0);
            }
        }
        // The next tile goes into the other LDS buffer NOW (last read one iteration ago, a barrier since): this wait covers the
        // prefetch loads only -- the stores below are issued after it and complete under the next tile's MFMAs (s_waitcnt vmcnt
        // counts loads and stores together: with the stores first, every tile waited for its own HBM write-backs: 40 % of the
        // wave cycles were parked there, SQ_WAIT_ANY).
        if (next < ntiles) stash(buf ^ 1, next * R);
        // ---- epilogue straight from the accumulator: lane l holds, for g = 0..3, the four CONSECUTIVE output columns
        //      32 cb + 8 g + 4 (l / 32) + {0..3} of point 32 pt + l % 32 -> one 16-B store per g and output array.  The eight
        //      16-B pieces of a 128-B line come from the same wave within a few instructions (L2 merges them); no LDS round
        //      trip, no barrier: the waves of a workgroup only meet at the X' tile.
        if (active) {
            const int n = n0 + 32 * pt + j;
            if (n < N) {
                const float ro = (!BWD && act == 2 && row_offset) ? row_offset[n] : 0.0f;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int c = 32 * cb + 8 * g + 4 * h;
                    if (c >= CO) continue;
                    float y[4] = { acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3] };
                    if (!BWD) {
                        float hh[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            y[u] += bv[4 * g + u];
                            if (act == 1) sgl_gelu_both(y[u], hh[u], y[u]);           // h and aux = gelu'(z) share one erf
                            else hh[u] = sgl_act(act, y[u], ro);
                        }
                        if ((CO & 3) == 0) {
                            if (out1) *(float4 *)(out1 + (size_t)n * CO + c) = make_float4(y[0], y[1], y[2], y[3]);    // aux
                            *(float4 *)(out0 + (size_t)n * CO + c) = make_float4(hh[0], hh[1], hh[2], hh[3]);           // h
                        } else {
#pragma unroll
                            for (int u = 0; u < 4; u++)
                                if (c + u < CO) { if (out1) out1[(size_t)n * CO + c + u] = y[u]; out0[(size_t)n * CO + c + u] = hh[u]; }
                        }
                    } else {
                        if ((CO & 3) == 0) *(float4 *)(out0 + (size_t)n * CO + c) = make_float4(y[0], y[1], y[2], y[3]);  // dx
                        else {
#pragma unroll
                            for (int u = 0; u < 4; u++) if (c + u < CO) out0[(size_t)n * CO + c + u] = y[u];
                        }
                    }
                }
            }
        }
        __syncthreads();                                                   // every wave has read its B operands of this tile;
    }                                                                      // the next tile's are in place
}


// ---- wide outputs (CO a multiple of 32: the trunk layers and every input-gradient product) ------------------------------------
// Same tile loop, two changes that the phase timings asked for (tools/linear_bench.py + phase-ablation builds, 128 -> 128 at 150 k
// points: MFMAs alone 41-58 us, epilogue + stores alone 43 us, loads + stores alone 55 us, the kernel 107 us = their SUM):
//  * the points are the A operand and M the B operand, so the accumulator holds output column 32 cb + j of 16 points and register
//    r leaves as two 128-B row segments per store instruction -- the full-rate store shape of a 32 x 32 accumulator
//    (MI355X_MICROARCH.md "access shape") instead of 32-B pieces of 32 rows;
//  * the epilogue of tile t - 1 (bias, GELU and gelu', stores) is issued BETWEEN the MFMAs of tile t: a wave issues in order, and a
//    dependent MFMA chain leaves ~60 idle issue cycles per instruction that the previous tile's element-wise work fills; the stores
//    are buffer stores whose range check drops the rows beyond N and a missing aux array (no branches inside the MFMA block).
template <int NQ> struct sgl_wide_cfg {
    static constexpr int CKP = NQ * 8, XS = CKP + 4;
    static constexpr int FIT = 6656 / (32 * XS);                           // 32-point tiles in 26 KiB (x 2 buffers x 3 workgroups per CU)
    static constexpr int PTCAP = FIT >= 4 ? 4 : (FIT >= 2 ? 2 : 1);
    static constexpr int PER = (32 * PTCAP * CKP / 4 + 255) / 256;         // float4 per thread and tile
};
__host__ __device__ inline int sgl_wide_point_tiles(int TO, int ptcap) { const int pt = 4 / TO; return pt < ptcap ? pt : ptcap; }

// Backward of a FAN (several layers reading the same activation, decoders.py:41-49, 75-94): the narrow heads beside a wide layer
// (xyz_offsets 3 + rotations 6 beside scales[0] 128; opacity 1 beside shs 48) used to ADD their dx in a pass of their own each --
// read-modify-write of the whole [N,Cin] gradient for a K = 1..6 product, 17-50 us per head on the step's critical chain.  Here
// their dz columns are 8 EXQ extra columns of the wide layer's reduction: X' = [dz_main | dz_side0 | dz_side1 | 0], M = [W_main; W_side0;
// W_side1; 0]^T.  The extra columns of a tile (R rows x 8 EXQ) are fetched by element (two look-ahead sets like the rest, every load a
// buffer load whose range check stands in for "not this source"), multiplied by sigmoid' where the head has one (opacity), stored
// as that head's dz for its weight gradient, and written into the LDS tile behind the main columns.
struct SgLinExtra {
    const float *dh[2], *aux[2], *W[2];       // [N,c], [N,c] | NULL, [c,Cin]
    float *dz[2];                             // [N,c] | NULL
    int c[2], act[2];                         // columns (0: unused), 0 | 2 (sigmoid: aux = the forward's output)
};

template <int NQ, bool BWD, bool GELU, bool VEC, int EXQ = 0>
__global__ void __launch_bounds__(256, (BWD || NQ == 12) ? 2 : 3)   // two register sets of look-ahead next to the weight registers: backward (x' and aux) and
                                                                    // the 96-input forward (64-point tiles) run two waves per SIMD, no spills
sg_linear_wide_kernel(int N, int CK, int CO, int act, const float *__restrict__ X, const float *__restrict__ Z,
                      const float *__restrict__ Mw, int m_co_stride, int m_ck_stride, const float *__restrict__ bias,
                      float *__restrict__ out0, float *__restrict__ out1, float *__restrict__ dz_out, int accum, SgLinExtra ex)
{
    static_assert(EXQ == 0 || BWD, "extra reduction columns: backward only");
    using cfg = sgl_wide_cfg<NQ + EXQ>;
    constexpr int CKP = cfg::CKP, HALF = CKP / 2, XS = cfg::XS, PER = cfg::PER;
    constexpr int NQT = NQ + EXQ, CKM = NQ * 8, EXC = EXQ * 8;             // float4 steps of the whole reduction; main columns (padded), extra columns
    constexpr int EPT = EXQ ? (32 * cfg::PTCAP * EXC + 255) / 256 : 1;     // extra elements per thread and tile
    extern __shared__ float sXd[];                                         // [2][R][XS]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int TO = CO >> 5;                                                // 1..4
    const int PT = sgl_wide_point_tiles(TO, cfg::PTCAP);
    const int cb = wave % TO, pt = wave / TO;
    const bool active = pt < PT;
    const int R = 32 * PT, TS = R * XS;
    const int j = lane & 31, h = lane >> 5;
    const int oc = 32 * cb + j;
    // this wave's 32 columns of M as the B operand: a[s] = M(oc, h HALF + s)
    float a[HALF];
    if (VEC && m_ck_stride == 1 && (m_co_stride & 3) == 0) {
        // forward orientation (M = W, row-major): a lane's registers are consecutive in memory -> 16-B loads.  As 4-B loads (64 per
        // lane, every one of them 64 different cache lines per wave) the 768 workgroups spent ~15 us of a 128 -> 128 layer fetching W.
#pragma unroll
        for (int s = 0; s < HALF; s += 4) {
            const int k = h * HALF + s;
            const float4 w4 = (active && k < CK) ? *(const float4 *)(Mw + (size_t)oc * m_co_stride + k) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            a[s] = w4.x; a[s + 1] = w4.y; a[s + 2] = w4.z; a[s + 3] = w4.w;
        }
    } else {
#pragma unroll
        for (int s = 0; s < HALF; s++) {
            const int k = h * HALF + s;
            float w = (active && k < CK) ? Mw[(size_t)oc * m_co_stride + (size_t)k * m_ck_stride] : 0.0f;
            if (EXQ && k >= CKM && active) {                               // a head's row of W [c, Cin]: reduction index k - CKM
                const int e = k - CKM;
                if (e < ex.c[0]) w = ex.W[0][(size_t)e * CO + oc];
                else if (e < ex.c[0] + ex.c[1]) w = ex.W[1][(size_t)(e - ex.c[0]) * CO + oc];
            }
            a[s] = w;
        }
    }
    const float bv = (!BWD && bias && active) ? bias[oc] : 0.0f;
    // output arrays as buffers: rows >= N (the ragged last tile) and a missing array fall outside num_records and are dropped
    const unsigned obytes = (unsigned)N * (unsigned)CO * 4u;
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(out0, 0, out0 ? obytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(out1, 0, out1 ? obytes : 0u, 0x00020000);
    const int CO4 = CO * 4;
    const int voff0 = ((32 * pt + 4 * h) * CO + oc) * 4;

    // ---- the X' tile: TWO tiles of look-ahead in registers (sets E and O).  With one tile in flight per workgroup the kernel moved
    // in-flight bytes / latency = 12.6 MB / 4.3 us = 2.9 TB/s however the rest was arranged (PMC: every byte algorithmic, matrix cores
    // busy 39 %).  Every load and store of the loop is a buffer operation whose range check stands in for the bounds tests (rows
    // beyond N, a missing array), so the loop body has no branch around a memory instruction and hipcc can count its
    // s_waitcnt vmcnt(n) -- with a branch it falls back to vmcnt(0), i.e. to waiting for the look-ahead it just issued.
    const int f4_per_row = CKP / 4;
    const int nf4 = R * f4_per_row;
    const unsigned xbytes = (unsigned)N * (unsigned)CK * 4u;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void *)X, 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsz = __builtin_amdgcn_make_buffer_rsrc((void *)Z, 0, (BWD && act != 0 && Z) ? xbytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(dz_out, 0, (BWD && act != 0 && dz_out) ? xbytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(out0, 0, (BWD && accum) ? obytes : 0u, 0x00020000);   // old dx (accumulate)
    const int CK4 = CK * 4;
    int xoff[PER], lofs[PER];                                             // byte offset inside a tile (out of range: not this thread's), LDS float index
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const int f = tid + 256 * q, row = f / f4_per_row, c = 4 * (f - row * f4_per_row);
        xoff[q] = (f < nf4 && c < CK) ? (row * CK + c) * 4 : 0x7ffffff0;
        lofs[q] = (f < nf4 && (!EXQ || c < CKM)) ? row * XS + c : -1;      // (the extra columns are written by their own path)
    }
    // extra columns: element e of the tile's [R][EXC] block -> row e / EXC, column e % EXC -> source 0 | source 1 | nothing
    const __amdgpu_buffer_rsrc_t re0 = __builtin_amdgcn_make_buffer_rsrc((void *)ex.dh[0], 0, EXQ && ex.c[0] ? (unsigned)N * ex.c[0] * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t re1 = __builtin_amdgcn_make_buffer_rsrc((void *)ex.dh[1], 0, EXQ && ex.c[1] ? (unsigned)N * ex.c[1] * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra0 = __builtin_amdgcn_make_buffer_rsrc((void *)ex.aux[0], 0, EXQ && ex.act[0] && ex.aux[0] ? (unsigned)N * ex.c[0] * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra1 = __builtin_amdgcn_make_buffer_rsrc((void *)ex.aux[1], 0, EXQ && ex.act[1] && ex.aux[1] ? (unsigned)N * ex.c[1] * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd0 = __builtin_amdgcn_make_buffer_rsrc(ex.dz[0], 0, EXQ && ex.act[0] && ex.dz[0] ? (unsigned)N * ex.c[0] * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd1 = __builtin_amdgcn_make_buffer_rsrc(ex.dz[1], 0, EXQ && ex.act[1] && ex.dz[1] ? (unsigned)N * ex.c[1] * 4u : 0u, 0x00020000);
    int eo0[EPT], eo1[EPT], elds[EPT];
#pragma unroll
    for (int i = 0; i < EPT; i++) {
        const int e = tid + 256 * i, row = EXQ ? e / (EXC ? EXC : 1) : 0, col = EXQ ? e - row * EXC : 0;
        const bool in = EXQ && row < R;
        eo0[i] = (in && col < ex.c[0]) ? (row * ex.c[0] + col) * 4 : 0x7ffffff0;
        eo1[i] = (in && col >= ex.c[0] && col < ex.c[0] + ex.c[1]) ? (row * ex.c[1] + col - ex.c[0]) * 4 : 0x7ffffff0;
        elds[i] = in ? row * XS + CKM + col : -1;
    }
    constexpr bool vec = VEC;                                              // CK a multiple of 4 (compile time: no branch around a load)
    auto ld4 = [&](const __amdgpu_buffer_rsrc_t &rs, int off) {
        float4 v;
        if (vec) {
            const auto u = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
            v = make_float4(__uint_as_float(u[0]), __uint_as_float(u[1]), __uint_as_float(u[2]), __uint_as_float(u[3]));
        } else {                                                           // CK not a multiple of 4: element loads; the row end by the column test
            const int c = (off % CK4) >> 2;                                // (off < 2^31: a real offset)
            float e[4];
#pragma unroll
            for (int u = 0; u < 4; u++)
                e[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (off != 0x7ffffff0 && c + u < CK) ? off + 4 * u : 0x7ffffff0, 0, 0));
            v = make_float4(e[0], e[1], e[2], e[3]);
        }
        return v;
    };
    float4 xrE[PER], xrO[PER], zrE[BWD ? PER : 1], zrO[BWD ? PER : 1];
    struct ExRegs { float d0[EPT], d1[EPT], g0[EPT], g1[EPT]; } exE, exO;
    auto ldx = [&](const __amdgpu_buffer_rsrc_t &rs, int o, int base) {
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, o == 0x7ffffff0 ? 0x7ffffff0 : o + base, 0, 0));
    };
    auto fetch_ex = [&](ExRegs &er, int n0) {
        if (!EXQ) return;
        const int b0 = n0 * ex.c[0] * 4, b1 = n0 * ex.c[1] * 4;
#pragma unroll
        for (int i = 0; i < EPT; i++) {
            er.d0[i] = ldx(re0, eo0[i], b0); er.d1[i] = ldx(re1, eo1[i], b1);
            er.g0[i] = ldx(ra0, eo0[i], b0); er.g1[i] = ldx(ra1, eo1[i], b1);      // (no activation: zero-sized buffer, the load returns 0)
        }
    };
    auto stash_ex = [&](const ExRegs &er, int buf, int n0) {
        if (!EXQ) return;
        const int b0 = n0 * ex.c[0] * 4, b1 = n0 * ex.c[1] * 4;
#pragma unroll
        for (int i = 0; i < EPT; i++) {
            // sigmoid' = h (1 - h) where the head has an activation; no branch around the stores (a head without one: zero-sized dz buffer)
            const float v0 = er.d0[i] * (ex.act[0] ? er.g0[i] * (1.0f - er.g0[i]) : 1.0f);
            const float v1 = er.d1[i] * (ex.act[1] ? er.g1[i] * (1.0f - er.g1[i]) : 1.0f);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0), rd0, eo0[i] == 0x7ffffff0 ? 0x7ffffff0 : eo0[i] + b0, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1), rd1, eo1[i] == 0x7ffffff0 ? 0x7ffffff0 : eo1[i] + b1, 0, 0);
            if (elds[i] >= 0) sXd[buf * TS + elds[i]] = v0 + v1;           // (one of them is the 0 of an out-of-range load; padding: both)
        }
    };
    auto fetch = [&](float4 (&xr)[PER], float4 (&zr)[BWD ? PER : 1], int n0) {
        const int t0 = n0 * CK4;                                           // (n0 <= N + 2 grid tiles: the launch keeps this below 2^31)
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int off = xoff[q] == 0x7ffffff0 ? 0x7ffffff0 : xoff[q] + t0;
            xr[q] = ld4(rsx, off);
            if (BWD) zr[q] = ld4(rsz, off);
        }
    };
    auto stash = [&](const float4 (&xr)[PER], const float4 (&zr)[BWD ? PER : 1], int buf, int n0) {
        const int t0 = n0 * CK4;
#pragma unroll
        for (int q = 0; q < PER; q++) {
            float4 v = xr[q];
            if (BWD && act != 0) {
                const float4 zz = zr[q];
                v.x *= sgl_act_grad_aux(act, zz.x); v.y *= sgl_act_grad_aux(act, zz.y);
                v.z *= sgl_act_grad_aux(act, zz.z); v.w *= sgl_act_grad_aux(act, zz.w);
                const int off = xoff[q] == 0x7ffffff0 ? 0x7ffffff0 : xoff[q] + t0;
                if (vec) {
                    typedef unsigned sgl_u4 __attribute__((ext_vector_type(4)));
                    const sgl_u4 u = { __float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w) };
                    __builtin_amdgcn_raw_buffer_store_b128(u, rsd, off, 0, 0);                       // dz
                } else {
                    const float e[4] = { v.x, v.y, v.z, v.w };
                    const int c = (off % CK4) >> 2;
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(e[u]), rsd, (off != 0x7ffffff0 && c + u < CK) ? off + 4 * u : 0x7ffffff0, 0, 0);
                }
            }
            if (lofs[q] >= 0) *(float4 *)&sXd[buf * TS + lofs[q]] = v;
        }
    };
    // one output register of a finished tile: bias + activation (forward), two 128-B row segments per store
    auto emit = [&](const sg_v16f &ac, int r, int n0p) {
        const int vo = voff0 + (n0p + 8 * (r >> 2) + (r & 3)) * CO4;
        if (!BWD) {
            float y = ac[r] + bv, hh;
            if (GELU) sgl_gelu_both(y, hh, y);                              // h and aux = gelu'(z) from one erf
            else hh = sgl_act(act, y, 0.0f);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(hh), rs0, vo, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), rs1, vo, 0, 0);
        } else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ac[r]), rs0, vo, 0, 0);
    };
    // the MFMAs of one tile; EPI: with the previous tile's epilogue spread between them
    auto mma = [&](auto EPI, int buf, const sg_v16f &accp, int n0p) {
        sg_v16f acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.0f;
        const float *xrow = &sXd[buf * TS + (32 * pt + j) * XS + h * HALF];  // A operand: X'[point j][h HALF + s], inputs permuted
        float4 b = *(const float4 *)xrow;
#pragma unroll
        for (int q = 0; q < NQT; q++) {
            float4 bn = b;
            if (q + 1 < NQT) bn = *(const float4 *)(xrow + 4 * (q + 1));
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a[4 * q], acc, 0, 0, 0);
            if (decltype(EPI)::value) {
#pragma unroll
                for (int r = (q * 16) / NQT; r < ((q + 1) * 16) / NQT; r++) emit(accp, r, n0p);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a[4 * q + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a[4 * q + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a[4 * q + 3], acc, 0, 0, 0);
            b = bn;
        }
        return acc;
    };
    const int ntiles = (N + R - 1) / R, G = gridDim.x;
    int tile = blockIdx.x;                                                 // grid <= ntiles
    fetch(xrE, zrE, tile * R); fetch_ex(exE, tile * R);
    fetch(xrO, zrO, (tile + G) * R); fetch_ex(exO, (tile + G) * R);        // (beyond the last tile: out of range, zeros)
    stash(xrE, zrE, 0, tile * R); stash_ex(exE, 0, tile * R);
    __syncthreads();
    // Everything issued so far (the weights!) has landed: said HERE, once, or the compiler -- merging the loop entry with the back
    // edge -- keeps a conservative wait in front of the first MFMA of every tile.
    __builtin_amdgcn_s_waitcnt(0x0F70);                                    // vmcnt(0)
    sg_v16f accp;
#pragma unroll
    for (int r = 0; r < 16; r++) accp[r] = 0.0f;
    int n0p = N;                                                           // no finished tile yet
    // one tile: the loads of tile + 2 G go out, the tile in LDS buffer `buf` is multiplied, tile + G moves from registers to LDS
    auto round = [&](float4 (&xa)[PER], float4 (&za)[BWD ? PER : 1], ExRegs &ea, const float4 (&xb)[PER], const float4 (&zb)[BWD ? PER : 1],
                     const ExRegs &eb, int t, int buf) {
        fetch(xa, za, (t + 2 * G) * R); fetch_ex(ea, (t + 2 * G) * R);
        if (BWD) {
            // no element-wise work to hide: dx leaves as soon as the tile's MFMAs are done (the stores are asynchronous);
            // accumulate (a layer input with several consumers, dx += dz W): the old values are requested before the MFMAs
            if (active) {
                float old[16];
#pragma unroll
                for (int r = 0; r < 16; r++)
                    old[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsa, voff0 + (t * R + 8 * (r >> 2) + (r & 3)) * CO4, 0, 0));
                sg_v16f acc = mma(std::false_type{}, buf, accp, 0);
#pragma unroll
                for (int r = 0; r < 16; r++) acc[r] += old[r];
#pragma unroll
                for (int r = 0; r < 16; r++) emit(acc, r, t * R);
            }
            stash(xb, zb, buf ^ 1, (t + G) * R); stash_ex(eb, buf ^ 1, (t + G) * R);
            __syncthreads();
        } else {
            // (the first tile runs the same block with n0p = N: its "previous tile" stores fall outside the arrays and are dropped --
            // a separate epilogue-free block would be a second path with a different store count, and the compiler would wait at the
            // stash for the lower one)
            sg_v16f acc = accp;
            if (active) acc = mma(std::true_type{}, buf, accp, n0p);
            stash(xb, zb, buf ^ 1, (t + G) * R);                           // the other buffer: last read one iteration ago
            __syncthreads();
            accp = acc; n0p = t * R;
        }
    };
    for (; tile < ntiles; tile += 2 * G) {
        round(xrE, zrE, exE, xrO, zrO, exO, tile, 0);
        if (tile + G >= ntiles) break;                                     // (a break, not an `if` around the second half: the back edge keeps its counts)
        round(xrO, zrO, exO, xrE, zrE, exE, tile + G, 1);
    }
    if (!BWD && active) {
#pragma unroll
        for (int r = 0; r < 16; r++) emit(accp, r, n0p);
    }
}

// ---- skinny heads, forward (xyz offsets 128 -> 3, rotations -> 6, the scale head's last layer 128 -> 1, opacity 64 -> 1) ---------
// A 32-column MFMA tile of which 1-6 columns are used is a bad way to stream [N,Cin]: the narrow kernel above moved 2.6 TB/s on
// these (25-30 us per head at 150 k points, all on the training step's serial chain).  Here the product is what it is -- CO dot
// products per row: L = Cin / 8 lanes share a row (two 16-byte loads each: a row is two contiguous 16 L-byte segments per load
// instruction), the CO partial dots are summed over the L lanes with log2 L shuffles, U rows groups are in flight per wave.
template <int CO, int L>                      // L = 8 (Cin = 64) or 16 (Cin = 128)
__global__ void __launch_bounds__(256)
sg_linear_skinny_kernel(int N, int act, const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias,
                        const float *__restrict__ row_offset, float *__restrict__ out)
{
    constexpr int CIN = 8 * L, RPW = 64 / L, U = 4;                        // rows per wave and load round; rounds in flight
    const int lane = threadIdx.x & 63, sub = lane % L, rg = lane / L;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
    float w[CO][8];
#pragma unroll
    for (int c = 0; c < CO; c++) {
        const float4 a = *(const float4 *)(W + (size_t)c * CIN + 4 * sub), b = *(const float4 *)(W + (size_t)c * CIN + 4 * (sub + L));
        w[c][0] = a.x; w[c][1] = a.y; w[c][2] = a.z; w[c][3] = a.w; w[c][4] = b.x; w[c][5] = b.y; w[c][6] = b.z; w[c][7] = b.w;
    }
    float bv[CO];
#pragma unroll
    for (int c = 0; c < CO; c++) bv[c] = bias ? bias[c] : 0.0f;
    const unsigned xbytes = (unsigned)N * CIN * 4u;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void *)X, 0, xbytes, 0x00020000);     // rows beyond N read as zero
    for (int r0 = wave * (RPW * U); r0 < N; r0 += nwaves * (RPW * U)) {
        float4 xa[U], xb[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int off = ((r0 + u * RPW + rg) * CIN + 4 * sub) * 4;
            const auto va = __builtin_amdgcn_raw_buffer_load_b128(rsx, off, 0, 0), vb = __builtin_amdgcn_raw_buffer_load_b128(rsx, off + 16 * L, 0, 0);
            xa[u] = make_float4(__uint_as_float(va[0]), __uint_as_float(va[1]), __uint_as_float(va[2]), __uint_as_float(va[3]));
            xb[u] = make_float4(__uint_as_float(vb[0]), __uint_as_float(vb[1]), __uint_as_float(vb[2]), __uint_as_float(vb[3]));
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            float d[CO];
#pragma unroll
            for (int c = 0; c < CO; c++) {
                float t = xa[u].x * w[c][0];
                t = fmaf(xa[u].y, w[c][1], t); t = fmaf(xa[u].z, w[c][2], t); t = fmaf(xa[u].w, w[c][3], t);
                t = fmaf(xb[u].x, w[c][4], t); t = fmaf(xb[u].y, w[c][5], t); t = fmaf(xb[u].z, w[c][6], t); t = fmaf(xb[u].w, w[c][7], t);
#pragma unroll
                for (int o = L / 2; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
                d[c] = t;
            }
            const int n = r0 + u * RPW + rg;
            if (sub == 0 && n < N) {
                const float ro = (act == 2 && row_offset) ? row_offset[n] : 0.0f;
#pragma unroll
                for (int c = 0; c < CO; c++) out[(size_t)n * CO + c] = sgl_act(act, d[c] + bv[c], ro);
            }
        }
    }
}

template <bool BWD>
static int sg_linear_launch(int N, int CK, int CO, int act, const float *X, const float *Z, const float *Mw, int s_co, int s_ck,
                            const float *bias, const float *row_offset, float *out0, float *out1, hipStream_t st, int accum = 0,
                            const SgLinExtra *exp = nullptr)
{
    if (N <= 0) return 0;
    if (CK < 1 || CK > SGL_MAXC || CO < 1 || CO > SGL_MAXC) return 1;
    const int nq = (CK + 7) / 8;
    SgLinExtra ex = {};
    if (exp) ex = *exp;
    const int exc = ex.c[0] + ex.c[1];
    if (exc) {
        // the heads' columns ride in the wide backward kernel: (CK padded to 8) + 8 | 16 reduction columns
        if (!BWD || (CO & 31) || exc > 16 || nq > 16 || (CK & 3) || ((long long)N + 2 * 512 * 128) * 144 * 4 >= 0x7fffffffLL) return 3;
        const int exq = exc <= 8 ? 1 : 2;
#define SGL_FAN_GO(NQv, EXQv)                                                                                                \
        do {                                                                                                                \
            using cfg = sgl_wide_cfg<NQv + EXQv>;                                                                           \
            const int R = 32 * sgl_wide_point_tiles(CO >> 5, cfg::PTCAP), ntiles = (N + R - 1) / R;                         \
            const int grid = ntiles < 512 ? ntiles : 512;                                                                   \
            const size_t dyn = (size_t)2 * R * cfg::XS * sizeof(float);                                                     \
            hipLaunchKernelGGL((sg_linear_wide_kernel<NQv, true, false, true, EXQv>), dim3(grid), dim3(256), dyn, st, N, CK, CO, act, X, Z, \
                               Mw, s_co, s_ck, bias, out0, (float *)nullptr, out1, accum, ex);                              \
        } while (0)
#define SGL_FAN(NQv) do { if (exq == 1) SGL_FAN_GO(NQv, 1); else SGL_FAN_GO(NQv, 2); } while (0)
        if (nq <= 6) SGL_FAN(6);
        else if (nq <= 8) SGL_FAN(8);
        else if (nq <= 12) SGL_FAN(12);
        else SGL_FAN(16);
#undef SGL_FAN
#undef SGL_FAN_GO
        return 0;
    }
    // buffer descriptors and tile offsets are 32-bit byte quantities on BOTH sides: X / aux / dz rows are CK floats, outputs CO floats,
    // and the look-ahead fetches reach up to two rounds of tiles (2 x 512 workgroups x 128 rows) past N before the range check
    // drops them -- so the bound is on max(CK, CO); larger problems take the narrow kernel below (64-bit addressing)
    const long long cmax = CK > CO ? CK : CO;
    if ((CO & 31) == 0 && !(act == 2 && row_offset) && ((long long)N + 2 * 512 * 128) * cmax * 4 < 0x7fffffffLL) {
        // wide outputs: forward h -> out0, aux -> out1; backward dx -> out0, dz -> out1
#define SGL_WIDE_GO(NQv, B, G, o0, o1, dz, acc)                                                                              \
        do {                                                                                                                \
            if ((CK & 3) == 0)                                                                                              \
                hipLaunchKernelGGL((sg_linear_wide_kernel<NQv, B, G, true>), dim3(grid), dim3(256), dyn, st, N, CK, CO, act, X, Z, \
                                   Mw, s_co, s_ck, bias, o0, o1, dz, acc, ex);                                              \
            else                                                                                                            \
                hipLaunchKernelGGL((sg_linear_wide_kernel<NQv, B, G, false>), dim3(grid), dim3(256), dyn, st, N, CK, CO, act, X, Z, \
                                   Mw, s_co, s_ck, bias, o0, o1, dz, acc, ex);                                              \
        } while (0)
#define SGL_WIDE(NQv)                                                                                                      \
        do {                                                                                                                \
            using cfg = sgl_wide_cfg<NQv>;                                                                                  \
            const int R = 32 * sgl_wide_point_tiles(CO >> 5, cfg::PTCAP), ntiles = (N + R - 1) / R;                         \
            const int gmax = 512;            /* two workgroups per CU: every workgroup loads the whole of M first */       \
            const int grid = ntiles < gmax ? ntiles : gmax;                                                                 \
            const size_t dyn = (size_t)2 * R * cfg::XS * sizeof(float);                                                     \
            if (BWD) SGL_WIDE_GO(NQv, true, false, out0, (float *)nullptr, out1, accum);                                    \
            else if (act == 1) SGL_WIDE_GO(NQv, false, true, out0, out1, (float *)nullptr, 0);                              \
            else SGL_WIDE_GO(NQv, false, false, out0, out1, (float *)nullptr, 0);                                           \
        } while (0)
        if (nq <= 1) SGL_WIDE(1);
        else if (nq <= 4) SGL_WIDE(4);
        else if (nq <= 6) SGL_WIDE(6);
        else if (nq <= 8) SGL_WIDE(8);
        else if (nq <= 12) SGL_WIDE(12);
        else SGL_WIDE(16);
#undef SGL_WIDE
#undef SGL_WIDE_GO
        return 0;
    }
    if (accum) return 2;                                                    // only the wide kernel accumulates
    const int nqt = nq <= 1 ? 1 : (nq <= 4 ? 4 : (nq <= 6 ? 6 : (nq <= 8 ? 8 : (nq <= 12 ? 12 : 16))));
    const int TO = (CO + 31) / 32, PT = sgl_point_tiles(TO, nqt * 8), R = 32 * PT;
    const int ntiles = (N + R - 1) / R;
    const int grid = ntiles < 768 ? ntiles : 768;                           // 3 resident workgroups per CU, persistent over tiles
    const size_t dyn = 0;
#define SGL_GO(NQv)                                                                                                       \
    hipLaunchKernelGGL((sg_linear_kernel<NQv, BWD>), dim3(grid), dim3(256), dyn, st, N, CK, CO, act, X, Z, Mw, s_co, s_ck, \
                       bias, row_offset, out0, out1)
    if (nq <= 1) SGL_GO(1);
    else if (nq <= 4) SGL_GO(4);
    else if (nq <= 6) SGL_GO(6);
    else if (nq <= 8) SGL_GO(8);
    else if (nq <= 12) SGL_GO(12);
    else SGL_GO(16);
#undef SGL_GO
    return 0;
}

// forward: x [N,Cin], W [Cout,Cin], bias [Cout] | NULL -> h_out [N,Cout]; aux_out | NULL: what the backward needs besides h
// (act 1: gelu'(z); otherwise the pre-activation z)
int sg_launch_linear_fwd(int N, int Cin, int Cout, int act, const float *x, const float *W, const float *bias,
                         const float *row_offset, float *z_out, float *h_out, hipStream_t st)
{
    // skinny heads: a streaming dot-product kernel (no aux output: act 0 needs none, the sigmoid's aux is its output)
    if (N > 0 && Cout <= 6 && (Cin == 64 || Cin == 128) && (act == 0 || act == 2) && !z_out && (long long)N * Cin * 4 < 0x7fffffffLL) {
        const int rows_per_wg = 4 * (Cin == 128 ? 4 : 8) * 4;
        int grid = (N + rows_per_wg - 1) / rows_per_wg;
        if (grid > 2048) grid = 2048;
#define SGL_SK(COv)                                                                                                        \
        do {                                                                                                              \
            if (Cin == 128) hipLaunchKernelGGL((sg_linear_skinny_kernel<COv, 16>), dim3(grid), dim3(256), 0, st, N, act, x, W, bias, row_offset, h_out); \
            else hipLaunchKernelGGL((sg_linear_skinny_kernel<COv, 8>), dim3(grid), dim3(256), 0, st, N, act, x, W, bias, row_offset, h_out);            \
        } while (0)
        switch (Cout) { case 1: SGL_SK(1); break; case 2: SGL_SK(2); break; case 3: SGL_SK(3); break; case 4: SGL_SK(4); break;
                        case 5: SGL_SK(5); break; default: SGL_SK(6); break; }
#undef SGL_SK
        return 0;
    }
    return sg_linear_launch<false>(N, Cin, Cout, act, x, nullptr, W, Cin, 1, bias, row_offset, h_out, z_out, st);
}
// backward: dh, aux [N,Cout] (act 1: the forward's aux; act 2: h; act 3: z), W [Cout,Cin] -> dz_out [N,Cout] = dh * act'
// (may be NULL), dx_out [N,Cin] = dz W
int sg_launch_linear_bwd(int N, int Cin, int Cout, int act, const float *z, const float *row_offset, const float *dh,
                         const float *W, float *dz_out, float *dx_out, hipStream_t st, int accumulate)
{
    // Y = dx [N, CO = Cin], X' = dz [N, CK = Cout], M(co = input column, ck = output column) = W[ck][co]
    return sg_linear_launch<true>(N, Cout, Cin, act, dh, z, W, 1, Cin, nullptr, row_offset, dx_out, dz_out, st, accumulate);
}
// the same with up to two narrow heads' products added in the same pass (SgLinearSide of include/sings_hip.h)
int sg_launch_linear_bwd_fan(int N, int Cin, int Cout, int act, const float *z, const float *dh, const float *W, float *dz_out,
                             float *dx_out, const SgLinearSide *side, hipStream_t st)
{
    SgLinExtra ex = {};
    for (int i = 0; i < 2; i++) {
        ex.c[i] = side[i].cout; ex.act[i] = side[i].act; ex.dh[i] = side[i].dh; ex.aux[i] = side[i].aux; ex.W[i] = side[i].W;
        ex.dz[i] = side[i].dz_out;
    }
    return sg_linear_launch<true>(N, Cout, Cin, act, dh, z, W, 1, Cin, nullptr, nullptr, dx_out, dz_out, st, 0, &ex);
}
