// Attribute decode upstream of the render path: multi-resolution tri-plane features and the
// element-wise parts of the two decoder MLPs, forward and backward.
//
// Replaces
//   HexPlaneField.forward / interpolate_ms_features / grid_sample_wrapper   sings/rec/models/modules/hexplane.py:46-105,163-190
//     (per scale: three F.grid_sample(bilinear, border, align_corners=True) calls, product over planes, concat over scales)
//   bias + activation of every nn.Linear of GeometryDecoder / AppearanceDecoder   modules/decoders.py:16-110
//     (GELU(erf), Sigmoid(x + opacity_offset), log(exp(x) + 1))
// The GEMMs themselves (N x {96,128,64} by <= 128) stay library GEMMs (rocBLAS through the host).
//
// Tri-plane kernels are gather / scatter bound.  The reference keeps a plane as [feat][H][W]: the 32 features of a
// texel are H*W floats apart, 1152 scattered 4-B loads per point.  Here every plane is first transposed to
// texel-major [H][W][feat] (one 128-B line per texel, 33 MB for the shipped 64/128/256 config -> L2/MALL resident) and
// 32 lanes = 32 features handle one point: 36 coalesced 128-B reads per point forward.  Backward recomputes the
// interpolation per point, counting-sorts the points by texel cell (once per projection) and adds the w * dL/dinterp
// rows into a texel-major gradient buffer in SORTED order, one 128-B float-atomic row per corner and cell CHANGE (the
// only float atomics in this library; the reference's grid_sample backward uses one per corner and point, so plane
// gradients are reproducible to rounding, not bitwise, in both) and transposes it back to the reference layout.
#include "sg_common.h"

#define SG_TP_FEAT 32
#define SG_TP_MAXS 4

struct SgTpDev {                       // device-side view
    int n_scales;
    int res[SG_TP_MAXS][3];
    long long fm_off[SG_TP_MAXS][3];   // float offset of the texel-major copy of plane (s, c) from the base pointer the kernel gets
    long long gm_off[SG_TP_MAXS][3];   // the same for the gradient planes (backward).  Feature-minor planes (SgTriplane.feature_minor):
                                       // the base is plane (0, 0) / its gradient, the offsets are address differences (any sign)
    float a0[3], ascale[3];            // normalised = (p - a0) * ascale - 1
};
__constant__ int sg_comb[3][2] = { { 0, 1 }, { 0, 2 }, { 1, 2 } };   // itertools.combinations(range(3), 2)

// [feat][H][W] -> [H][W][feat]   (and the reverse for gradients); blockIdx.y = plane, ONE launch for all planes
struct SgTpPlanes {
    const float *src[SG_TP_MAXS * 3];
    float *dst[SG_TP_MAXS * 3];
    int HW[SG_TP_MAXS * 3];
};
__global__ void __launch_bounds__(256)
sg_plane_to_fm_kernel(SgTpPlanes P)
{
    __shared__ float t[32][33];
    const int HW = P.HW[blockIdx.y];
    const float *__restrict__ src = P.src[blockIdx.y];
    float *__restrict__ dst = P.dst[blockIdx.y];
    const int x0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 texels x 32 features per block
    if (x0 >= HW || !src || !dst) return;
    for (int f = ty; f < 32; f += 8) t[f][tx] = x0 + tx < HW ? src[(size_t)f * HW + x0 + tx] : 0.0f;
    __syncthreads();
    for (int p = ty; p < 32; p += 8)
        if (x0 + p < HW) dst[(size_t)(x0 + p) * 32 + tx] = t[tx][p];
}
__global__ void __launch_bounds__(256)
sg_plane_from_fm_kernel(SgTpPlanes P)
{
    __shared__ float t[32][33];
    const int HW = P.HW[blockIdx.y];
    const float *__restrict__ src = P.src[blockIdx.y];
    float *__restrict__ dst = P.dst[blockIdx.y];
    const int x0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    if (x0 >= HW || !src || !dst) return;
    for (int p = ty; p < 32; p += 8) t[p][tx] = x0 + p < HW ? src[(size_t)(x0 + p) * 32 + tx] : 0.0f;
    __syncthreads();
    for (int f = ty; f < 32; f += 8)
        if (x0 + tx < HW) dst[(size_t)f * HW + x0 + tx] = t[tx][f];
}

struct SgTexel {
    int x0, x1, y0, y1;
    float wx, wy;          // weight of x1 / y1
    float gx, gy;          // d(ix)/du, d(iy)/dv incl. the border-clip mask
};
// F.grid_sample coordinate rule: align_corners=True, padding_mode='border'
__device__ __forceinline__ void sg_texel(float u, float v, int W, int H, SgTexel &t)
{
    float ix = (u + 1.0f) * 0.5f * (float)(W - 1), iy = (v + 1.0f) * 0.5f * (float)(H - 1);
    // clip_coordinates_set_grad: the coordinate gradient is zero AT and beyond the border
    t.gx = (ix > 0.0f && ix < (float)(W - 1)) ? 0.5f * (float)(W - 1) : 0.0f;
    t.gy = (iy > 0.0f && iy < (float)(H - 1)) ? 0.5f * (float)(H - 1) : 0.0f;
    ix = fminf(fmaxf(ix, 0.0f), (float)(W - 1)); iy = fminf(fmaxf(iy, 0.0f), (float)(H - 1));
    const float fx = floorf(ix), fy = floorf(iy);
    t.x0 = (int)fx; t.y0 = (int)fy;
    t.x1 = min(t.x0 + 1, W - 1); t.y1 = min(t.y0 + 1, H - 1);
    t.wx = ix - fx; t.wy = iy - fy;
}

// 32 lanes per point (lane = feature), 8 points per 256-thread workgroup
__global__ void __launch_bounds__(256)
sg_triplane_fwd_kernel(SgTpDev d, int N, const float *__restrict__ xyz, const float *__restrict__ fm,
                       float *__restrict__ feats)
{
    const int n = blockIdx.x * 8 + (threadIdx.x >> 5), f = threadIdx.x & 31;
    if (n >= N) return;
    float p[3];
#pragma unroll
    for (int a = 0; a < 3; a++) p[a] = (xyz[3 * (size_t)n + a] - d.a0[a]) * d.ascale[a] - 1.0f;
    const int F = d.n_scales * SG_TP_FEAT;
    for (int s = 0; s < d.n_scales; s++) {
        float prod = 1.0f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int a = sg_comb[c][0], b = sg_comb[c][1];
            const int W = d.res[s][a], H = d.res[s][b];
            SgTexel t;
            sg_texel(p[a], p[b], W, H, t);
            const float *pl = fm + d.fm_off[s][c];
            const float v00 = pl[((size_t)t.y0 * W + t.x0) * 32 + f], v01 = pl[((size_t)t.y0 * W + t.x1) * 32 + f];
            const float v10 = pl[((size_t)t.y1 * W + t.x0) * 32 + f], v11 = pl[((size_t)t.y1 * W + t.x1) * 32 + f];
            // same weights as grid_sample: nw = (1-wx)(1-wy), ne = wx (1-wy), sw = (1-wx) wy, se = wx wy
            const float interp = v00 * ((1.0f - t.wx) * (1.0f - t.wy)) + v01 * (t.wx * (1.0f - t.wy)) +
                                 v10 * ((1.0f - t.wx) * t.wy) + v11 * (t.wx * t.wy);
            prod = prod * interp;
        }
        feats[(size_t)n * F + s * SG_TP_FEAT + f] = prod;
    }
}

// ---- plane gradient: sort, then segmented scatter ---------------------------------------------------------
// A plain scatter (one float atomic per texel corner and feature: 1.7e8 memory-side adds for 150k points) was the
// slowest kernel of a training step (1.2 ms; an avatar's points pile up on the same texels).  Instead:
//   1. sg_tp_cell_count / bsum / scan / pos   counting sort of the points by the texel cell of the FINEST level they
//                               fall in (+ parity sub-keys, see sg_tp_sort_key), once per projection (xy, xz, yz): 3 sorts
//                               serve all 9 planes;
//   2. sg_tp_bwd_point_kernel   per point: recompute the interpolation, dL/dxyz, and ONE 128-B row per plane
//                               dL/dfeat * (product of the other two planes) plus a 16-B (cell, weights) record, both
//                               stored at the point's SORTED slot of that plane (random 128-B writes are fire-and-forget);
//   3. sg_tp_sorted_scatter_kernel   a half-wave streams SG_TP_RUN consecutive slots (rows and records are sequential in memory)
//                               and keeps the corner sums of the current texel cell in registers; a row of float atomics
//                               leaves only when the cell changes: ~2.5e7 atomics instead of 1.7e8 on the avatar, which is
//                               what is left of the kernel's time (46 us of streaming + ~120 us at the ~0.8 TB/s the
//                               memory-side float atomics sustain).  Tried and dropped: a texel-owner gather (one
//                               workgroup per 8x8 texel block, LDS accumulators, no global atomics at all) -- an avatar
//                               fills ~2 % of the texels, so ~100 workgroups did all the work (19 ms); rows left in point
//                               order and gathered through the permutation (152 us of dependent random reads).
//   Avatar cloud, 150k points: 1 180 us (plain scatter) -> 533 us (32-point runs) -> 411 us (128-point runs: every run
//   boundary is a flush of four rows, and there are 9 planes x N / run of them).
// Sum order inside a texel is not fixed: reproducible to rounding, like the scatter and the reference.
struct SgTpSort {                      // per projection c: fine grid = max over the scales of the resolutions
    int Wf[3], Hf[3];
    int nsub[3], sub_scale[3][SG_TP_MAXS];   // coarser scales of the projection: 2 key bits each (cell parity in x, y)
    int nkeys[3];                      // Wf * Hf << (2 * nsub)
    size_t start_off[3];               // uint32 offset of start[c][0 .. nkeys] in the cell workspace
    size_t bsum_off[3];                // uint32 offset of the 1024-key block sums
};
// Sort key of a point for projection c: its cell on the finest grid, then -- because align_corners cells of different
// levels are NOT nested (63 / 127 / 255 cells across) -- the parity of its cell on every coarser level: a fine cell
// overlaps at most two coarser cells per axis, so inside one fine cell the points come out grouped by the cell of every
// level.  (Without the parity bits the points of a fine cell that a coarse boundary crosses alternate between the two
// coarse cells in arrival order: 60 000 cell changes per plane instead of 12 000 on the avatar.)
__device__ __forceinline__ uint32_t sg_tp_sort_key(const SgTpDev &d, const SgTpSort &g, int c, const float *__restrict__ xyz, int n)
{
    const int a = sg_comb[c][0], b = sg_comb[c][1];
    const float u = (xyz[3 * (size_t)n + a] - d.a0[a]) * d.ascale[a] - 1.0f;
    const float v = (xyz[3 * (size_t)n + b] - d.a0[b]) * d.ascale[b] - 1.0f;
    SgTexel t;
    sg_texel(u, v, g.Wf[c], g.Hf[c], t);
    uint32_t key = (uint32_t)(t.y0 * g.Wf[c] + t.x0);
    for (int i = 0; i < g.nsub[c]; i++) {
        const int s = g.sub_scale[c][i];
        sg_texel(u, v, d.res[s][a], d.res[s][b], t);
        key = (key << 2) | (uint32_t)((t.x0 & 1) | ((t.y0 & 1) << 1));
    }
    return key;
}
__global__ void __launch_bounds__(256)
sg_tp_cell_count_kernel(SgTpDev d, SgTpSort g, int N, const float *__restrict__ xyz, uint32_t *__restrict__ count,
                        uint32_t *__restrict__ keys, uint32_t *__restrict__ rank)
{
    const int n = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (n >= N) return;
    const uint32_t key = sg_tp_sort_key(d, g, c, xyz, n);
    keys[(size_t)c * N + n] = key;
    rank[(size_t)c * N + n] = atomicAdd(&count[g.start_off[c] + key], 1u);
}
// exclusive scan of the key counters in two launches: sums of 1024-key blocks, then every block adds up the block sums
// in front of it (<= nkeys / 1024 words) and scans its own keys.  start[i] = number of points with key < i, i in [0, nkeys].
__global__ void __launch_bounds__(256)
sg_tp_cell_bsum_kernel(SgTpSort g, const uint32_t *__restrict__ count, uint32_t *__restrict__ bsum)
{
    __shared__ uint32_t sW[4];
    const int c = blockIdx.y, nkeys = g.nkeys[c];
    const int i0 = blockIdx.x * 1024 + threadIdx.x * 4;
    if ((int)blockIdx.x * 1024 > nkeys) return;
    const uint32_t *cnt = count + g.start_off[c];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) sum += i0 + k < nkeys ? cnt[i0 + k] : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if ((threadIdx.x & 63) == 0) sW[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) bsum[g.bsum_off[c] + blockIdx.x] = sW[0] + sW[1] + sW[2] + sW[3];
}
__global__ void __launch_bounds__(256)
sg_tp_cell_scan_kernel(SgTpSort g, const uint32_t *__restrict__ count, const uint32_t *__restrict__ bsum,
                       uint32_t *__restrict__ start)
{
    __shared__ uint32_t sW[4];
    __shared__ uint32_t sPre;
    const int c = blockIdx.y, nkeys = g.nkeys[c];
    const int base = blockIdx.x * 1024;
    if (base > nkeys) return;
    const uint32_t *cnt = count + g.start_off[c];
    uint32_t pre = 0;
    for (int i = threadIdx.x; i < (int)blockIdx.x; i += 256) pre += bsum[g.bsum_off[c] + i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pre += __shfl_xor(pre, o, 64);
    if (lane == 0) sW[wave] = pre;
    __syncthreads();
    if (threadIdx.x == 0) sPre = sW[0] + sW[1] + sW[2] + sW[3];
    __syncthreads();
    const int i0 = base + threadIdx.x * 4;
    uint32_t v[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { v[k] = i0 + k < nkeys ? cnt[i0 + k] : 0u; sum += v[k]; }
    uint32_t incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
    __syncthreads();
    if (lane == 63) sW[wave] = incl;
    __syncthreads();
    uint32_t ex = sPre + incl - sum;
    for (int w = 0; w < wave; w++) ex += sW[w];
    uint32_t *st = start + g.start_off[c];
#pragma unroll
    for (int k = 0; k < 4; k++) { if (i0 + k <= nkeys) st[i0 + k] = ex; ex += v[k]; }
}
// pos[c][n] = position of point n in the sorted order of projection c
__global__ void __launch_bounds__(256)
sg_tp_cell_pos_kernel(SgTpSort g, int N, const uint32_t *__restrict__ start, const uint32_t *__restrict__ keys,
                      uint32_t *__restrict__ rank_pos)
{
    const int n = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (n >= N) return;
    const uint32_t k = start[g.start_off[c] + keys[(size_t)c * N + n]] + rank_pos[(size_t)c * N + n];
    rank_pos[(size_t)c * N + n] = k < (uint32_t)N ? k : (uint32_t)N - 1u;
}

// per point (32 lanes = 32 features): G rows + dL/dxyz.  Same arithmetic as the scatter kernel above.
__global__ void __launch_bounds__(256)
sg_tp_bwd_point_kernel(SgTpDev d, int N, const float *__restrict__ xyz, const float *__restrict__ fm,
                       const float *__restrict__ dfeats, const uint32_t *__restrict__ pos, float *__restrict__ G,
                       float4 *__restrict__ cellrec, float *__restrict__ dxyz)
{
    const int n = blockIdx.x * 8 + (threadIdx.x >> 5), f = threadIdx.x & 31;
    if (n >= N) return;
    float p[3];
#pragma unroll
    for (int a = 0; a < 3; a++) p[a] = (xyz[3 * (size_t)n + a] - d.a0[a]) * d.ascale[a] - 1.0f;
    const int F = d.n_scales * SG_TP_FEAT;
    float dp[3] = { 0.0f, 0.0f, 0.0f };
    for (int s = 0; s < d.n_scales; s++) {
        SgTexel t[3];
        float v[3][4], interp[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int a = sg_comb[c][0], b = sg_comb[c][1];
            const int W = d.res[s][a], H = d.res[s][b];
            sg_texel(p[a], p[b], W, H, t[c]);
            const float *pl = fm + d.fm_off[s][c];
            v[c][0] = pl[((size_t)t[c].y0 * W + t[c].x0) * 32 + f]; v[c][1] = pl[((size_t)t[c].y0 * W + t[c].x1) * 32 + f];
            v[c][2] = pl[((size_t)t[c].y1 * W + t[c].x0) * 32 + f]; v[c][3] = pl[((size_t)t[c].y1 * W + t[c].x1) * 32 + f];
            interp[c] = v[c][0] * ((1.0f - t[c].wx) * (1.0f - t[c].wy)) + v[c][1] * (t[c].wx * (1.0f - t[c].wy)) +
                        v[c][2] * ((1.0f - t[c].wx) * t[c].wy) + v[c][3] * (t[c].wx * t[c].wy);
        }
        const float g = dfeats[(size_t)n * F + s * SG_TP_FEAT + f];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int a = sg_comb[c][0], b = sg_comb[c][1];
            const int W = d.res[s][a];
            const float gi = g * interp[(c + 1) % 3] * interp[(c + 2) % 3];      // dL/d interp_c
            // the row and its (cell, weights) record go to the point's slot in the SORTED order of projection c: the
            // scatter kernel then streams both (random 128-B writes here are fire-and-forget; random reads there were not)
            const size_t slot = (size_t)(s * 3 + c) * N + pos[(size_t)c * N + n];
            G[slot * 32 + f] = gi;
            const float wx = t[c].wx, wy = t[c].wy;
            if (f == 0)
                cellrec[slot] = make_float4(__uint_as_float((uint32_t)(t[c].y0 * W + t[c].x0)),
                                            __uint_as_float((uint32_t)(t[c].y1 * W + t[c].x1)), wx, wy);
            dp[a] += gi * ((v[c][1] - v[c][0]) * (1.0f - wy) + (v[c][3] - v[c][2]) * wy) * t[c].gx;
            dp[b] += gi * ((v[c][2] - v[c][0]) * (1.0f - wx) + (v[c][3] - v[c][1]) * wx) * t[c].gy;
        }
    }
    if (dxyz) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            float r = dp[a];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) r += __shfl_xor(r, o, 64);
            if (f == 0) dxyz[3 * (size_t)n + a] = r * d.ascale[a];
        }
    }
}

#define SG_TP_RUN 128                  // consecutive sorted points per half-wave
// blockIdx.y = plane (s, c).  Every half-wave (32 lanes = 32 features) walks SG_TP_RUN consecutive slots of the
// projection's sorted order -- G rows and (cell, weights) records are sequential in memory -- and keeps the four
// corner sums of the CURRENT texel cell in registers; they leave through one row of float atomics per corner only when
// the cell changes.  Consecutive points share the cell of every level most of the time: the avatar's 1.7e8 atomics
// become a few million, and the work is balanced by construction (points, not texels, are dealt out).
__global__ void __launch_bounds__(256)
sg_tp_sorted_scatter_kernel(SgTpDev d, int N, const float *__restrict__ G, const float4 *__restrict__ cellrec,
                            float *__restrict__ gfm)
{
    const int sc = blockIdx.y, s = sc / 3, c = sc % 3;
    const int f = threadIdx.x & 31;
    const int k0 = (blockIdx.x * 8 + (threadIdx.x >> 5)) * SG_TP_RUN;
    if (k0 >= N) return;
    const int k1 = min(k0 + SG_TP_RUN, N);
    const float *Gp = G + (size_t)sc * N * 32;
    const float4 *rp = cellrec + (size_t)sc * N;
    float *gp = gfm + d.gm_off[s][c];
    const int W = d.res[s][sg_comb[c][0]];
    uint32_t c00 = 0xffffffffu, c11 = 0;
    float acc00 = 0.0f, acc01 = 0.0f, acc10 = 0.0f, acc11 = 0.0f;
    // eight slots per round, the NEXT round's sixteen loads in flight while this one is accumulated (indices clamped, not tested:
    // no load behind a branch); without the look-ahead every round of eight points started with a full memory latency
    float4 r[8], rn[8];
    float gi[8], gn[8];
    auto fetch = [&](float4 (&rr)[8], float (&gg)[8], int k) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int kk = min(k + j, k1 - 1);
            rr[j] = rp[kk];
            gg[j] = Gp[(size_t)kk * 32 + f];
        }
    };
    fetch(r, gi, k0);
    for (int k = k0; k < k1; k += 8) {
        fetch(rn, gn, min(k + 8, k1 - 1));
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (k + j >= k1) break;
            const uint32_t n00 = __float_as_uint(r[j].x), n11 = __float_as_uint(r[j].y);
            if (n00 != c00) {
                if (c00 != 0xffffffffu) {
                    // corners: (y0,x0) = c00, (y1,x1) = c11, (y0,x1) = c00 + dx, (y1,x0) = c11 - dx, dx = x1 - x0
                    const uint32_t dx = (c11 - c00) % (uint32_t)W;
                    unsafeAtomicAdd(&gp[(size_t)c00 * 32 + f], acc00);
                    unsafeAtomicAdd(&gp[(size_t)(c00 + dx) * 32 + f], acc01);
                    unsafeAtomicAdd(&gp[(size_t)(c11 - dx) * 32 + f], acc10);
                    unsafeAtomicAdd(&gp[(size_t)c11 * 32 + f], acc11);
                }
                c00 = n00; c11 = n11;
                acc00 = acc01 = acc10 = acc11 = 0.0f;
            }
            const float wx = r[j].z, wy = r[j].w;
            acc00 += gi[j] * ((1.0f - wx) * (1.0f - wy));
            acc01 += gi[j] * (wx * (1.0f - wy));
            acc10 += gi[j] * ((1.0f - wx) * wy);
            acc11 += gi[j] * (wx * wy);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) { r[j] = rn[j]; gi[j] = gn[j]; }
    }
    if (c00 != 0xffffffffu) {
        const uint32_t dx = (c11 - c00) % (uint32_t)W;
        unsafeAtomicAdd(&gp[(size_t)c00 * 32 + f], acc00);
        unsafeAtomicAdd(&gp[(size_t)(c00 + dx) * 32 + f], acc01);
        unsafeAtomicAdd(&gp[(size_t)(c11 - dx) * 32 + f], acc10);
        unsafeAtomicAdd(&gp[(size_t)c11 * 32 + f], acc11);
    }
}

// ---- bias + activation ------------------------------------------------------------------------------------
// act: 0 identity, 1 GELU (erf), 2 sigmoid(z + row_offset), 3 log(exp(z) + 1)
__device__ __forceinline__ float sg_gelu(float z) { return 0.5f * z * (1.0f + erff(z * 0.70710678118654752440f)); }
__device__ __forceinline__ float sg_gelu_grad(float z)
{
    const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752440f));
    return cdf + z * 0.39894228040143267794f * __expf(-0.5f * z * z);
}
__global__ void __launch_bounds__(256)
sg_bias_act_fwd_kernel(size_t total, int C, int act, const float *__restrict__ y, const float *__restrict__ bias,
                       const float *__restrict__ row_offset, float *__restrict__ z_out, float *__restrict__ h_out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % (size_t)C);
    float z = y[i] + (bias ? bias[c] : 0.0f);
    if (z_out) z_out[i] = z;
    float h = z;
    if (act == 1) h = sg_gelu(z);
    else if (act == 2) { const float t = z + (row_offset ? row_offset[i / (size_t)C] : 0.0f); h = 1.0f / (1.0f + expf(-t)); }
    else if (act == 3) h = logf(expf(z) + 1.0f);
    h_out[i] = h;
}
// dz = dh * act'(z); column sums of dz (bias gradient) as per-block partials [gridDim.x][C] (C <= 128)
__global__ void __launch_bounds__(256)
sg_bias_act_bwd_kernel(int N, int C, int act, int rows_per_block, const float *__restrict__ z,
                       const float *__restrict__ row_offset, const float *__restrict__ dh, float *__restrict__ dz,
                       float *__restrict__ partial)
{
    __shared__ float sAcc[256];
    // thread t handles column t % Cp and row phase t / Cp, Cp = C rounded up to a divisor-friendly width
    const int Cp = C <= 1 ? 1 : (C <= 4 ? 4 : (C <= 64 ? 64 : 128));
    const int col = threadIdx.x % Cp, ph = threadIdx.x / Cp, nph = 256 / Cp;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, N);
    float acc = 0.0f;
    if (col < C)
        for (int r = r0 + ph; r < r1; r += nph) {
            const size_t i = (size_t)r * C + col;
            const float zz = z[i], g = dh[i];
            float d = g;
            if (act == 1) d = g * sg_gelu_grad(zz);
            else if (act == 2) { const float s = 1.0f / (1.0f + expf(-(zz + (row_offset ? row_offset[r] : 0.0f)))); d = g * s * (1.0f - s); }
            else if (act == 3) { const float e = expf(zz); d = g * e / (e + 1.0f); }
            dz[i] = d;
            acc += d;
        }
    sAcc[threadIdx.x] = acc;
    __syncthreads();
    if (ph == 0 && col < C) {
        float t = 0.0f;
        for (int q = 0; q < nph; q++) t += sAcc[q * Cp + col];
        partial[(size_t)blockIdx.x * C + col] = t;
    }
}
// dz = dh * act'(z) only (the bias gradient comes out of sg_weight_grad): plain streaming kernel
__global__ void __launch_bounds__(256)
sg_act_bwd_kernel(size_t total, int C, int act, const float *__restrict__ z, const float *__restrict__ row_offset,
                  const float *__restrict__ dh, float *__restrict__ dz)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const float zz = z[i], g = dh[i];
    float d = g;
    if (act == 1) d = g * sg_gelu_grad(zz);
    else if (act == 2) { const float s = 1.0f / (1.0f + expf(-(zz + (row_offset ? row_offset[i / (size_t)C] : 0.0f)))); d = g * s * (1.0f - s); }
    else if (act == 3) { const float e = expf(zz); d = g * e / (e + 1.0f); }
    dz[i] = d;
}
__global__ void __launch_bounds__(128)
sg_colsum_reduce_kernel(const float *__restrict__ partial, int nblocks, int C, float *__restrict__ out)
{
    const int c = blockIdx.x * 128 + threadIdx.x;
    if (c >= C) return;
    double t = 0.0;
    for (int b = 0; b < nblocks; b++) t += (double)partial[(size_t)b * C + c];
    out[c] = (float)t;
}

// ---- host side ----------------------------------------------------------------------------------------------
static size_t sg_tp_layout(const SgTriplane *tp, SgTpDev *d)
{
    size_t o = 0;
    d->n_scales = tp->n_scales;
    for (int s = 0; s < tp->n_scales; s++)
        for (int c = 0; c < 3; c++) {
            const int comb[3][2] = { { 0, 1 }, { 0, 2 }, { 1, 2 } };
            d->fm_off[s][c] = (long long)o; d->gm_off[s][c] = (long long)o;
            o += (size_t)tp->res[s][comb[c][0]] * tp->res[s][comb[c][1]] * SG_TP_FEAT;
            o = (o + 63) & ~(size_t)63;
        }
    for (int s = 0; s < tp->n_scales; s++)
        for (int a = 0; a < 3; a++) d->res[s][a] = tp->res[s][a];
    for (int a = 0; a < 3; a++) {
        d->a0[a] = tp->aabb[0][a];
        d->ascale[a] = 2.0f / (tp->aabb[1][a] - tp->aabb[0][a]);          // hexplane.py:161 normalize_aabb
    }
    return o;
}
int sg_tp_check(const SgTriplane *tp)
{
    if (!tp || tp->n_scales < 1 || tp->n_scales > SG_TP_MAXS || tp->feat != SG_TP_FEAT) return 1;
    for (int s = 0; s < tp->n_scales; s++)
        for (int a = 0; a < 3; a++)
            if (tp->res[s][a] < 2) return 1;
    return 0;
}
size_t sg_triplane_ws_bytes_impl(const SgTriplane *tp)
{
    SgTpDev d;
    return 2 * sg_align(sg_tp_layout(tp, &d) * 4);           // texel-major planes + texel-major gradients
}
static void sg_tp_upload(const SgTriplane *tp, const SgTpDev &d, float *fm, hipStream_t st)
{
    const int comb[3][2] = { { 0, 1 }, { 0, 2 }, { 1, 2 } };
    SgTpPlanes P;
    int maxHW = 0;
    for (int s = 0; s < tp->n_scales; s++)
        for (int c = 0; c < 3; c++) {
            const int HW = tp->res[s][comb[c][0]] * tp->res[s][comb[c][1]];
            P.src[s * 3 + c] = tp->planes[s][c]; P.dst[s * 3 + c] = fm + d.fm_off[s][c]; P.HW[s * 3 + c] = HW;
            maxHW = HW > maxHW ? HW : maxHW;
        }
    hipLaunchKernelGGL(sg_plane_to_fm_kernel, dim3((maxHW + 31) / 32, tp->n_scales * 3), dim3(256), 0, st, P);
}
// feature-minor planes ([1, F, H, W] tensors in channels_last memory format = [H][W][F]): the kernels read them in place
static const float *sg_tp_direct(const SgTriplane *tp, SgTpDev *d)
{
    const float *base = tp->planes[0][0];
    for (int s = 0; s < tp->n_scales; s++)
        for (int c = 0; c < 3; c++) d->fm_off[s][c] = (long long)(tp->planes[s][c] - base);
    return base;
}
void sg_launch_triplane_fwd(const SgTriplane *tp, int N, const float *xyz, void *ws, float *feats, hipStream_t st)
{
    SgTpDev d;
    sg_tp_layout(tp, &d);
    const float *fm = (const float *)ws;
    if (tp->feature_minor) fm = sg_tp_direct(tp, &d);        // the planes ARE texel-major: no copy
    else sg_tp_upload(tp, d, (float *)ws, st);
    hipLaunchKernelGGL(sg_triplane_fwd_kernel, dim3((N + 7) / 8), dim3(256), 0, st, d, N, xyz, fm, feats);
}
static void sg_tp_sort_layout(const SgTriplane *tp, SgTpSort *g, size_t *cell_words, size_t *bsum_words)
{
    const int comb[3][2] = { { 0, 1 }, { 0, 2 }, { 1, 2 } };
    size_t o = 0, ob = 0;
    for (int c = 0; c < 3; c++) {
        int wf = 2, hf = 2;
        for (int s = 0; s < tp->n_scales; s++) {
            wf = tp->res[s][comb[c][0]] > wf ? tp->res[s][comb[c][0]] : wf;
            hf = tp->res[s][comb[c][1]] > hf ? tp->res[s][comb[c][1]] : hf;
        }
        g->Wf[c] = wf; g->Hf[c] = hf;
        g->nsub[c] = 0;
        for (int s = 0; s < tp->n_scales; s++)
            if (tp->res[s][comb[c][0]] != wf || tp->res[s][comb[c][1]] != hf) g->sub_scale[c][g->nsub[c]++] = s;
        // key space capped at 2^24: past that the coarsest sub-keys are dropped (still a valid sort, more cell changes)
        while (g->nsub[c] > 0 && ((size_t)wf * hf << (2 * g->nsub[c])) > ((size_t)1 << 24)) g->nsub[c]--;
        g->nkeys[c] = (int)((size_t)wf * hf << (2 * g->nsub[c]));
        g->start_off[c] = o;
        o += ((size_t)g->nkeys[c] + 1 + 63) & ~(size_t)63;
        g->bsum_off[c] = ob;
        ob += ((size_t)g->nkeys[c] / 1024 + 1 + 63) & ~(size_t)63;
    }
    *cell_words = o; *bsum_words = ob;
}
// backward workspace: texel-major planes | texel-major gradients | G rows [S*3][N][32] | count | start | block sums |
// keys | rank / sorted position | (cell, weights) records [S*3][N]
size_t sg_triplane_bwd_ws_bytes_impl(const SgTriplane *tp, int N)
{
    SgTpDev d; SgTpSort g; size_t cw, bw;
    const size_t floats = sg_tp_layout(tp, &d), n = N > 0 ? N : 1;
    sg_tp_sort_layout(tp, &g, &cw, &bw);
    return 2 * sg_align(floats * 4) + sg_align((size_t)tp->n_scales * 3 * n * 32 * 4) + 2 * sg_align(cw * 4) +
           sg_align(bw * 4) + 2 * sg_align(3 * n * 4) + sg_align((size_t)tp->n_scales * 3 * n * 16);
}
// The backward in two halves.  `prepare` needs only the points and the planes -- texel-major copy of the planes, zeroed gradient
// planes, the three counting sorts -- and can run any time after the forward (e.g. on a side stream under the decoders: it is a
// quarter of the backward and pure latency); `run` needs dL/dfeats.
struct SgTpBwdWs { float *fm, *gfm, *G; uint32_t *count, *start, *bsum, *keys, *rank; float4 *cellrec; };
static size_t sg_tp_bwd_carve(const SgTriplane *tp, int N, void *ws, SgTpDev *d, SgTpSort *g, SgTpBwdWs *w, size_t *cw_out)
{
    size_t cw, bw;
    const size_t floats = sg_tp_layout(tp, d), n = N;
    sg_tp_sort_layout(tp, g, &cw, &bw);
    char *b = (char *)ws;
    w->fm = (float *)b; b += sg_align(floats * 4);
    w->gfm = (float *)b; b += sg_align(floats * 4);
    w->G = (float *)b; b += sg_align((size_t)tp->n_scales * 3 * n * 32 * 4);
    w->count = (uint32_t *)b; b += sg_align(cw * 4);
    w->start = (uint32_t *)b; b += sg_align(cw * 4);
    w->bsum = (uint32_t *)b; b += sg_align(bw * 4);
    w->keys = (uint32_t *)b; b += sg_align(3 * n * 4);
    w->rank = (uint32_t *)b; b += sg_align(3 * n * 4);          // arrival rank inside the key, then sorted position
    w->cellrec = (float4 *)b;
    *cw_out = cw;
    return floats;
}
int sg_launch_triplane_bwd_prepare(const SgTriplane *tp, int N, const float *xyz, void *ws, hipStream_t st)
{
    SgTpDev d; SgTpSort g; SgTpBwdWs w; size_t cw;
    const size_t floats = sg_tp_bwd_carve(tp, N, ws, &d, &g, &w, &cw);
    if (!tp->feature_minor) sg_tp_upload(tp, d, w.fm, st);    // (parameters may have changed since the forward call)
    sg_zero_async(w.count, cw * 4, st);
    const int nb = (N + 255) / 256;
    int max_keys = 0;
    for (int c = 0; c < 3; c++) max_keys = g.nkeys[c] > max_keys ? g.nkeys[c] : max_keys;
    if (!tp->feature_minor) sg_zero_async(w.gfm, floats * 4, st);      // (feature-minor: the caller's gradient planes, zeroed by the caller)
    const int nkb = max_keys / 1024 + 1;
    hipLaunchKernelGGL(sg_tp_cell_count_kernel, dim3(nb, 3), dim3(256), 0, st, d, g, N, xyz, w.count, w.keys, w.rank);
    hipLaunchKernelGGL(sg_tp_cell_bsum_kernel, dim3(nkb, 3), dim3(256), 0, st, g, w.count, w.bsum);
    hipLaunchKernelGGL(sg_tp_cell_scan_kernel, dim3(nkb, 3), dim3(256), 0, st, g, w.count, w.bsum, w.start);
    hipLaunchKernelGGL(sg_tp_cell_pos_kernel, dim3(nb, 3), dim3(256), 0, st, g, N, w.start, w.keys, w.rank);
    return 0;
}
int sg_launch_triplane_bwd_run(const SgTriplane *tp, int N, const float *xyz, void *ws, const float *dfeats,
                               float *const dplanes[SG_TP_MAXS][3], float *dxyz, hipStream_t st)
{
    SgTpDev d; SgTpSort g; SgTpBwdWs w; size_t cw;
    sg_tp_bwd_carve(tp, N, ws, &d, &g, &w, &cw);
    const int comb[3][2] = { { 0, 1 }, { 0, 2 }, { 1, 2 } };
    const float *fm = w.fm;
    float *gfm = w.gfm;
    if (tp->feature_minor) {
        // planes and gradient planes in place: no texel-major copies (the caller zero-filled dplanes)
        fm = sg_tp_direct(tp, &d);
        gfm = dplanes[0][0];
        for (int s = 0; s < tp->n_scales; s++)
            for (int c = 0; c < 3; c++) d.gm_off[s][c] = (long long)(dplanes[s][c] - gfm);
    }
    hipLaunchKernelGGL(sg_tp_bwd_point_kernel, dim3((N + 7) / 8), dim3(256), 0, st, d, N, xyz, fm, dfeats, w.rank, w.G, w.cellrec,
                       dxyz);
    hipLaunchKernelGGL(sg_tp_sorted_scatter_kernel, dim3((N + 8 * SG_TP_RUN - 1) / (8 * SG_TP_RUN), tp->n_scales * 3),
                       dim3(256), 0, st, d, N, w.G, w.cellrec, gfm);
    if (tp->feature_minor) return 0;
    SgTpPlanes P;
    int maxHW = 0;
    for (int s = 0; s < tp->n_scales; s++)
        for (int c = 0; c < 3; c++) {
            const int HW = tp->res[s][comb[c][0]] * tp->res[s][comb[c][1]];
            P.src[s * 3 + c] = w.gfm + d.fm_off[s][c]; P.dst[s * 3 + c] = dplanes[s][c]; P.HW[s * 3 + c] = HW;
            maxHW = HW > maxHW ? HW : maxHW;
        }
    hipLaunchKernelGGL(sg_plane_from_fm_kernel, dim3((maxHW + 31) / 32, tp->n_scales * 3), dim3(256), 0, st, P);
    return 0;
}
int sg_launch_triplane_bwd(const SgTriplane *tp, int N, const float *xyz, void *ws, const float *dfeats,
                           float *const dplanes[SG_TP_MAXS][3], float *dxyz, hipStream_t st)
{
    sg_launch_triplane_bwd_prepare(tp, N, xyz, ws, st);
    return sg_launch_triplane_bwd_run(tp, N, xyz, ws, dfeats, dplanes, dxyz, st);
}

void sg_launch_bias_act_fwd(int N, int C, int act, const float *y, const float *bias, const float *row_offset,
                            float *z_out, float *h_out, hipStream_t st)
{
    const size_t total = (size_t)N * C;
    hipLaunchKernelGGL(sg_bias_act_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, total, C, act, y,
                       bias, row_offset, z_out, h_out);
}
static inline int sg_ba_blocks(int N) { int b = (N + 511) / 512; return b < 1 ? 1 : (b > 1024 ? 1024 : b); }
size_t sg_bias_act_ws_bytes_impl(int N, int C) { return sg_align((size_t)sg_ba_blocks(N) * C * 4); }
void sg_launch_bias_act_bwd(int N, int C, int act, const float *z, const float *row_offset, const float *dh, void *ws,
                            float *dz, float *dbias, hipStream_t st)
{
    if (!dbias) {
        const size_t total = (size_t)N * C;
        hipLaunchKernelGGL(sg_act_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, total, C, act, z,
                           row_offset, dh, dz);
        return;
    }
    const int nb = sg_ba_blocks(N), rpb = (N + nb - 1) / nb;
    float *partial = (float *)ws;
    hipLaunchKernelGGL(sg_bias_act_bwd_kernel, dim3(nb), dim3(256), 0, st, N, C, act, rpb, z, row_offset, dh, dz, partial);
    if (dbias) hipLaunchKernelGGL(sg_colsum_reduce_kernel, dim3((C + 127) / 128), dim3(128), 0, st, partial, nb, C, dbias);
}

// ---- the isotropic scale head's tail (decoders.py:88-94 + sings_hybrid.py:286-292): scales = log(exp(z) + 1) and scales_aux = z,
// both repeated to three columns -- ONE launch each way instead of an activation kernel and two repeats forward, a row-sum reduction,
// an activation-gradient kernel and two additions backward (each 5-9 us on the training step's serial chain).
__global__ void __launch_bounds__(256)
sg_scales_head_fwd_kernel(int N, const float *__restrict__ z, float *__restrict__ scales, float *__restrict__ aux)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float zz = z[n], h = logf(expf(zz) + 1.0f);
    scales[3 * (size_t)n] = h; scales[3 * (size_t)n + 1] = h; scales[3 * (size_t)n + 2] = h;
    aux[3 * (size_t)n] = zz; aux[3 * (size_t)n + 1] = zz; aux[3 * (size_t)n + 2] = zz;
}
// dz = (sum_c dscales[n,c]) e^z / (e^z + 1) + sum_c daux[n,c]   (either gradient may be missing)
__global__ void __launch_bounds__(256)
sg_scales_head_bwd_kernel(int N, const float *__restrict__ z, const float *__restrict__ dscales, const float *__restrict__ daux,
                          float *__restrict__ dz)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float d = 0.0f;
    if (dscales) {
        const float e = expf(z[n]);
        d = ((dscales[3 * (size_t)n] + dscales[3 * (size_t)n + 1]) + dscales[3 * (size_t)n + 2]) * (e / (e + 1.0f));
    }
    if (daux) d += (daux[3 * (size_t)n] + daux[3 * (size_t)n + 1]) + daux[3 * (size_t)n + 2];
    dz[n] = d;
}
void sg_launch_scales_head(int N, const float *z, float *scales, float *aux, const float *dscales, const float *daux, float *dz,
                           hipStream_t st)
{
    if (dz) hipLaunchKernelGGL(sg_scales_head_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, st, N, z, dscales, daux, dz);
    else hipLaunchKernelGGL(sg_scales_head_fwd_kernel, dim3((N + 255) / 256), dim3(256), 0, st, N, z, scales, aux);
}

// ---- weight / bias gradient of a decoder layer: dW [Cout,Cin] = dz^T x, db [Cout] = column sums of dz ---------------
// A GEMM whose reduction dimension is the N ~ 10^5 points and whose output is at most 128 x 128: the library kernels
// chosen for this shape ran at 330-430 us per layer (150 k points).  Here a workgroup owns a slice of rows and the whole
// output: wave w accumulates the 32 output rows [32w, 32w+32) x all Cin columns on the matrix cores
// (v_mfma_f32_32x32x2_f32, exact fp32 products, k-ordered accumulation).  The MFMA operand layout IS the memory layout
// of two consecutive rows (lane l: A = dz[n + l/32][32w + l%32], B = x[n + l/32][32t + l%32]), so operands are loaded
// straight from global memory with 128-B coalesced half-wave reads -- no LDS, 1 + Cin/32 loads per Cin/32 MFMAs.
// Per-workgroup partials are summed in a fixed order by sg_wgrad_reduce_kernel (deterministic).
typedef float sg_v16f __attribute__((ext_vector_type(16)));
#define SG_WG_ROWS 256

// RR = rows of x / dz per round (one barrier per round): 32 for <= 96 input columns, 16 for 128 (64 rows lose again).
// Nine decoder layers at 150k points: 604 us (16-row rounds, one round of look-ahead, 256-row slices dealt round-robin)
// -> 550 (round size) -> 484 (two rounds of look-ahead) -> 423 us (one equally long row range per workgroup).
// Round 2: every load is a BUFFER load on a descriptor of the workgroup's own row range -- rows beyond it read as zero, so the
// loop has no bounds branches at all and the compiler can count its s_waitcnt vmcnt(n) (with a branch per load it fell back to
// vmcnt(0) right behind the look-ahead loads, which therefore never overlapped anything).
// SPLIT (cout_pad <= 64: the 64- and 48-output layers): only two waves own output rows, so the four waves form two PAIRS that share a
// round's RR rows -- pair g multiplies rows [g RR/2, (g + 1) RR/2) -- and leave TWO partial slabs per workgroup (the reduce kernel
// sums 2 x as many): 43.5 -> ~25 us for dW [64 x 96] at 150 k points, where two of the four matrix cores of a CU used to idle.
template <int TI, int RR, bool SPLIT>
__global__ void __launch_bounds__(256)
sg_wgrad_kernel(int N, int Cout, int Cin, const float *__restrict__ dz, const float *__restrict__ x,
                float *__restrict__ partial, float *__restrict__ bpartial, int cout_pad, int chunk)
{
    // the four waves need the same RR rows of x per round: they go through LDS once (double-buffered, one barrier per
    // round) instead of being read from L2/HBM by every wave (4x the traffic made the direct version bandwidth-bound)
    __shared__ float sX[2][RR][TI * 32 + 4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wv = SPLIT ? (wave & 1) : wave, grp = SPLIT ? (wave >> 1) : 0;           // output-row block, row group of the round
    constexpr int NU = SPLIT ? RR / 4 : RR / 2;                                        // row pairs per wave and round
    const int u0 = grp * NU;
    const bool active = 32 * wv < cout_pad;
    const int o = 32 * wv + (lane & 31), half = lane >> 5;
    constexpr int CIN = TI * 32;
    constexpr int F4 = RR * CIN / 4;          // float4 elements of one slab
    static_assert(F4 % 256 == 0, "a slab is a whole number of float4 per thread");
    constexpr int PER = F4 / 256;
    sg_v16f acc[TI];
#pragma unroll
    for (int t = 0; t < TI; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.0f;
    float bsum = 0.0f;
    (void)Cin;
    // every workgroup takes ONE contiguous range of `chunk` rows (the last one may be short)
    const int r0 = blockIdx.x * chunk, len = min(chunk, N - r0);
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void *)(x + (size_t)r0 * CIN), 0, len * CIN * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsz = __builtin_amdgcn_make_buffer_rsrc((void *)(dz + (size_t)r0 * Cout), 0, len * Cout * 4, 0x00020000);
    const int zoff = (active && o < Cout) ? (half * Cout + o) * 4 : 0x7ffffff0;   // columns beyond Cout: out of range = 0
    int xoff[PER];
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const int f = threadIdx.x + 256 * q, row = f / (CIN / 4), c4 = f - row * (CIN / 4);
        xoff[q] = (row * CIN + 4 * c4) * 4;
    }
    // two register sets (E, O): while round n computes out of LDS, the operands of round n + 1 sit in one set (stashed
    // to the other LDS buffer at the end of the round) and the loads of round n + 2 are in flight into the other --
    // a round's MFMAs (0.85 us) are shorter than a memory round trip, one round of look-ahead left the waves waiting
    float4 xrE[PER], xrO[PER];
    float anE[NU], anO[NU];
    auto fetch = [&](float4 (&xr)[PER], float (&an)[NU], int n) {
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsx, xoff[q] + n * (CIN * 4), 0, 0);
            xr[q] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
#pragma unroll
        for (int u = 0; u < NU; u++)
            an[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsz, zoff + (n + 2 * (u0 + u)) * (Cout * 4), 0, 0));
    };
    auto stash = [&](const float4 (&xr)[PER], int buf) {
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int f = threadIdx.x + 256 * q, row = f / (CIN / 4), c4 = f - row * (CIN / 4);
            *(float4 *)&sX[buf][row][4 * c4] = xr[q];
        }
    };
    // round n: a-values from `cur` (its x rows are in sX[buf]); `oth` holds round n + 1
    auto round = [&](float4 (&xc)[PER], float (&ac)[NU], const float4 (&xo)[PER], int n, int buf) {
        float a[NU];
#pragma unroll
        for (int u = 0; u < NU; u++) a[u] = ac[u];
        fetch(xc, ac, n + 2 * RR);
        if (active) {
#pragma unroll
            for (int u = 0; u < NU; u++) {
                bsum += a[u];
#pragma unroll
                for (int t = 0; t < TI; t++)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], sX[buf][2 * (u0 + u) + half][32 * t + (lane & 31)], acc[t], 0, 0, 0);
            }
        }
        stash(xo, buf ^ 1);
        __syncthreads();
    };
    if (len > 0) {
        fetch(xrE, anE, 0);
        fetch(xrO, anO, RR);
        stash(xrE, 0);
        __syncthreads();
        for (int n = 0; n < len; n += 2 * RR) {
            round(xrE, anE, xrO, n, 0);
            if (n + RR >= len) break;                  // (a break, not an `if` around the second round: the back edge keeps its static load count)
            round(xrO, anO, xrE, n + RR, 1);
        }
    }
    if (!active) return;
    // D layout of the 32x32 tile: lane l, register r -> row 8 * (r / 4) + 4 * (l / 32) + r % 4, column l % 32
    const size_t slab = SPLIT ? (size_t)blockIdx.x * 2 + grp : (size_t)blockIdx.x;      // (SPLIT: one slab per wave pair)
    float *pw = partial + slab * cout_pad * Cin;
#pragma unroll
    for (int t = 0; t < TI; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int i = 32 * wv + 8 * (r >> 2) + 4 * half + (r & 3);
            pw[(size_t)i * Cin + 32 * t + (lane & 31)] = acc[t][r];
        }
    bsum += __shfl_xor(bsum, 32, 64);
    if (half == 0) bpartial[slab * cout_pad + o] = bsum;
}

// Heads with <= 12 outputs (xyz offsets, rotations, scale, opacity): the product is a handful of dot products per input column --
// a streaming read of x (HBM-bound) instead of a 32-row MFMA tile of which 1-3 rows are used (80 us for 128 inputs).
// Thread = one float4 of input columns x one of 256 / (Cin / 4) row phases; fixed-order tree over the phases in LDS.
template <int CO>
__global__ void __launch_bounds__(256)
sg_wgrad_small_kernel(int N, int Cin, const float *__restrict__ dz, const float *__restrict__ x,
                      float *__restrict__ partial, float *__restrict__ bpartial)
{
    __shared__ float4 sAcc[CO][256];
    __shared__ float sB[CO][256];
    const int c4n = Cin >> 2, c4 = threadIdx.x % c4n, ph = threadIdx.x / c4n, nph = 256 / c4n;
    float4 acc[CO];
    float bs[CO];
#pragma unroll
    for (int c = 0; c < CO; c++) { acc[c] = make_float4(0, 0, 0, 0); bs[c] = 0.0f; }
    if (ph < nph) {
        const int stride = gridDim.x * nph;
        int r = blockIdx.x * nph + ph;
        for (; r + 3 * stride < N; r += 4 * stride) {
            float4 xv[4];
            float g[4][CO];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                xv[u] = *(const float4 *)(x + (size_t)(r + u * stride) * Cin + 4 * c4);
#pragma unroll
                for (int c = 0; c < CO; c++) g[u][c] = dz[(size_t)(r + u * stride) * CO + c];
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int c = 0; c < CO; c++) {
                    acc[c].x += g[u][c] * xv[u].x; acc[c].y += g[u][c] * xv[u].y;
                    acc[c].z += g[u][c] * xv[u].z; acc[c].w += g[u][c] * xv[u].w;
                    bs[c] += g[u][c];
                }
        }
        for (; r < N; r += stride) {
            const float4 xv = *(const float4 *)(x + (size_t)r * Cin + 4 * c4);
#pragma unroll
            for (int c = 0; c < CO; c++) {
                const float g = dz[(size_t)r * CO + c];
                acc[c].x += g * xv.x; acc[c].y += g * xv.y; acc[c].z += g * xv.z; acc[c].w += g * xv.w;
                bs[c] += g;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CO; c++) { sAcc[c][threadIdx.x] = acc[c]; sB[c][threadIdx.x] = bs[c]; }
    __syncthreads();
    if (ph == 0) {
#pragma unroll
        for (int c = 0; c < CO; c++) {
            float4 t = make_float4(0, 0, 0, 0);
            for (int q = 0; q < nph; q++) {
                const float4 v = sAcc[c][q * c4n + c4];
                t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            }
            *(float4 *)(partial + ((size_t)blockIdx.x * CO + c) * Cin + 4 * c4) = t;
            if (c4 == 0) {                                     // every phase saw the same dz rows only once: column 0's copy
                float b = 0.0f;
                for (int q = 0; q < nph; q++) b += sB[c][q * c4n];
                bpartial[(size_t)blockIdx.x * CO + c] = b;
            }
        }
    }
}

// 16 output elements x 16 slices of the per-workgroup partials per 256-thread workgroup (fixed slices, fixed order):
// the sum over <= 256 partials is latency-bound, so it is spread over many short chains
__global__ void __launch_bounds__(256)
sg_wgrad_reduce_kernel(const float *__restrict__ partial, const float *__restrict__ bpartial, int nwg, int Cout, int Cin,
                       int cout_pad, float *__restrict__ dW, float *__restrict__ db)
{
    __shared__ float sS[16][17];
    const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    const int total = Cout * Cin;
    const int per = (nwg + 15) / 16, w0 = sl * per, w1 = min(w0 + per, nwg);
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    // sixteen partials per round, every load issued before the first addition (clamped indices, zero weights beyond the slice): with
    // <= 256 workgroups a slice IS one round -- as a 4-at-a-time loop it was four dependent memory round trips, the whole kernel
    if (e < total || (db && e < total + Cout)) {
        const bool isw = e < total;
        const int o = isw ? e / Cin : e - total, c = isw ? e - o * Cin : 0;
        const float *p = isw ? partial + (size_t)o * Cin + c : bpartial + o;
        const size_t st = isw ? (size_t)cout_pad * Cin : (size_t)cout_pad;
        for (int w = w0; w < w1; w += 16) {
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int wi = w + i < w1 ? w + i : w1 - 1;
                v[i] = p[(size_t)wi * st];
            }
#pragma unroll
            for (int i = 0; i < 16; i += 4) {
                s0 += w + i < w1 ? v[i] : 0.0f; s1 += w + i + 1 < w1 ? v[i + 1] : 0.0f;
                s2 += w + i + 2 < w1 ? v[i + 2] : 0.0f; s3 += w + i + 3 < w1 ? v[i + 3] : 0.0f;
            }
        }
    }
    sS[sl][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0) {
        float t = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; q++) t += sS[q][el];
        if (e < total) dW[e] = t;
        else if (db && e < total + Cout) db[e - total] = t;
    }
}

// MFMA kernel: ONE workgroup per CU (256 equal row ranges; a range need not be a whole number of rounds -- the rows beyond it read
// as zero).  A wave's TI independent accumulator chains keep its SIMD's matrix core busy on their own, and the partials -- one
// Cout x Cin slab per workgroup, written once and read once by the reduce kernel -- are what more workgroups multiply: 768 / 512 /
// 256 workgroups: 76 / 65 / 59 us for a 128 x 128 layer at 150 k points, of which the reduce is 15 / 11 / 6.
static inline int sg_wg_chunk(int N, int rr)
{
    (void)rr;
    const int c = (N + 255) / 256;
    return c < 32 ? 32 : c;
}
static inline int sg_wg_count(int N) { const int n = (N + SG_WG_ROWS - 1) / SG_WG_ROWS; return n < 512 ? n : 512; }
static inline int sg_wg_count_max(int N) { const int n = 2 * ((N + 31) / 32); return n < 768 ? n : 768; }   // bound over the kernels' slab counts (SPLIT: two per workgroup)
size_t sg_weight_grad_ws_bytes_impl(int N, int Cout, int Cin)
{
    const size_t cp = (size_t)((Cout + 31) / 32) * 32;
    return sg_align((size_t)sg_wg_count_max(N) * cp * Cin * 4) + sg_align((size_t)sg_wg_count_max(N) * cp * 4);
}
int sg_launch_weight_grad(int N, int Cout, int Cin, const float *dz, const float *x, void *ws, float *dW, float *db,
                          hipStream_t st)
{
    if (Cin % 32 != 0 || Cin > 128 || Cout > 128 || Cout < 1) return 1;
    int cp = ((Cout + 31) / 32) * 32;
    int nwg = sg_wg_count(N);
    const int ti = Cin / 32;
    float *partial = (float *)ws;
    float *bpartial = (float *)((char *)ws + sg_align((size_t)sg_wg_count_max(N) * cp * Cin * 4));
    if (Cout <= 12) {
        // (partials are [nwg][Cout][Cin] here: the reduce kernel takes the row pitch as a parameter)
#define SG_WGS(C) case C: hipLaunchKernelGGL(sg_wgrad_small_kernel<C>, dim3(nwg), dim3(256), 0, st, N, Cin, dz, x, partial, bpartial); break
        switch (Cout) {
            SG_WGS(1); SG_WGS(2); SG_WGS(3); SG_WGS(4); SG_WGS(5); SG_WGS(6); SG_WGS(7); SG_WGS(8); SG_WGS(9); SG_WGS(10); SG_WGS(11);
            default: hipLaunchKernelGGL(sg_wgrad_small_kernel<12>, dim3(nwg), dim3(256), 0, st, N, Cin, dz, x, partial, bpartial); break;
        }
#undef SG_WGS
        cp = Cout;
    } else {
        const int rr = ti == 4 ? 16 : 32, chunk = sg_wg_chunk(N, rr);
        nwg = (N + chunk - 1) / chunk;
        const bool split = cp <= 64;                          // two waves own output rows: the other pair shares the round's rows
#define SG_WGK(T)                                                                                                                   \
        do {                                                                                                                        \
            if (split) hipLaunchKernelGGL((sg_wgrad_kernel<T, (T == 4 ? 16 : 32), true>), dim3(nwg), dim3(256), 0, st, N, Cout, Cin, dz, x, partial, bpartial, cp, chunk); \
            else hipLaunchKernelGGL((sg_wgrad_kernel<T, (T == 4 ? 16 : 32), false>), dim3(nwg), dim3(256), 0, st, N, Cout, Cin, dz, x, partial, bpartial, cp, chunk);    \
        } while (0)
        switch (ti) { case 1: SG_WGK(1); break; case 2: SG_WGK(2); break; case 3: SG_WGK(3); break; default: SG_WGK(4); break; }
#undef SG_WGK
        if (split) nwg *= 2;                                  // (slabs for the reduce)
    }
    hipLaunchKernelGGL(sg_wgrad_reduce_kernel, dim3((Cout * Cin + Cout + 15) / 16), dim3(256), 0, st, partial, bpartial, nwg,
                       Cout, Cin, cp, dW, db);
    return 0;
}
