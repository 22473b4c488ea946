// Per-Gaussian projection math shared by the plain and the LBS-fused kernels.
// Forward: SURVEY.md App. A.1; backward: App. A.5 (reference call site
// sings/rec/renderer/gs_renderer_single.py:87-95).  Exact operation order -- see sg_math.h.
#pragma once
#include "sg_math.h"

template <int D>
__device__ __forceinline__ void sg_eval_sh(const float *__restrict__ sh, const float dir[3],
                                           float rgb[3], uint32_t &clampbits)
{
    float b[16];
    sg_sh_basis<D>(dir, b);
    constexpr int nc = (D + 1) * (D + 1);
    clampbits = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float acc = b[0] * sh[c];
#pragma unroll
        for (int k = 1; k < nc; k++) acc = acc + b[k] * sh[3 * k + c];
        acc = acc + 0.5f;
        if (acc < 0.0f) { clampbits |= 1u << c; acc = 0.0f; }
        rgb[c] = acc;
    }
}


struct SgProj {
    int mr;                      // radius in pixels (0: culled / invisible)
    uint32_t tt, clampbits;      // tiles touched, SH clamp mask
    int x0, y0, x1, y1;          // tile rectangle
    float pix[2], conic[3], rgb[3], depth;
};

// p: world-space mean, s3: scales, q: (r,x,y,z) un-normalised quaternion; c6pre / colpre: optional
// precomputed covariance (6) / colour (3) of THIS Gaussian; sh: this Gaussian's [M,3] SH rows.
template <int D>
__device__ __forceinline__ void sg_project_fwd(const SgCam &c, const float p[3], const float s3[3], const float q[4],
                                               const float *__restrict__ c6pre, const float *__restrict__ colpre,
                                               const float *__restrict__ sh, SgProj &o)
{
    o.mr = 0; o.tt = 0; o.clampbits = 0; o.x0 = o.y0 = o.x1 = o.y1 = 0;
    o.pix[0] = o.pix[1] = 0; o.conic[0] = o.conic[1] = o.conic[2] = 0; o.rgb[0] = o.rgb[1] = o.rgb[2] = 0; o.depth = 0;
    float pv[3];
    sg_xf4x3(p, c.view, pv);
    if (!(pv[2] > 0.2f)) return;
    float ph[4];
    sg_xf4x4(p, c.proj, ph);
    float pw = 1.0f / (ph[3] + 0.0000001f);
    float ppx = ph[0] * pw, ppy = ph[1] * pw;
    float c6[6];
    if (c6pre) {
#pragma unroll
        for (int k = 0; k < 6; k++) c6[k] = c6pre[k];
    } else {
        sg_cov3d(s3, c.mod, q, c6);
    }
    float Mm[6], tc[3], abc[3]; bool xin, yin;
    sg_proj_jac(pv, c.fx, c.fy, c.tanfovx, c.tanfovy, c.view, Mm, tc, xin, yin);
    sg_cov2d(Mm, c6, abc);
    float det = abc[0] * abc[2] - abc[1] * abc[1];
    if (det == 0.0f) return;
    float det_inv = 1.0f / det;
    float conic[3] = { abc[2] * det_inv, -abc[1] * det_inv, abc[0] * det_inv };
    float mid = 0.5f * (abc[0] + abc[2]);
    float dd = mid * mid - det; dd = dd < 0.1f ? 0.1f : dd;
    float sq = sqrtf(dd);
    float l1 = mid + sq, l2 = mid - sq;
    float lm = l1 > l2 ? l1 : l2;
    float my_radius = ceilf(3.0f * sqrtf(lm));
    // A radius that is not a positive number below 2^30 pixels (NaN / Inf covariance: non-finite scales or rotations): culled.
    // Upstream converts it with (int) -- 0 for NaN on its hardware --, stores radii = 0 and emits no keys for it: nothing is
    // rendered there either; here the Gaussian also takes no pair slot and a zero gradient (tests/test_gpu_degenerate.py).
    if (!(my_radius > 0.0f && my_radius < 1073741824.0f)) return;
    float pix0 = ((ppx + 1.0f) * (float)c.W - 1.0f) * 0.5f;
    float pix1 = ((ppy + 1.0f) * (float)c.H - 1.0f) * 0.5f;
    int r = (int)my_radius, x0, y0, x1, y1;
    sg_rect(pix0, pix1, r, c.gx, c.gy, x0, y0, x1, y1);
    uint32_t area = (uint32_t)((x1 - x0) * (y1 - y0));
    if (area == 0) return;
    o.mr = r; o.tt = area; o.depth = pv[2];
    o.x0 = x0; o.y0 = y0; o.x1 = x1; o.y1 = y1;
    o.pix[0] = pix0; o.pix[1] = pix1; o.conic[0] = conic[0]; o.conic[1] = conic[1]; o.conic[2] = conic[2];
    if (colpre) {
        o.rgb[0] = colpre[0]; o.rgb[1] = colpre[1]; o.rgb[2] = colpre[2];
    } else {
        float dir[3] = { p[0] - c.campos[0], p[1] - c.campos[1], p[2] - c.campos[2] };
        float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
        dir[0] = dir[0] / len; dir[1] = dir[1] / len; dir[2] = dir[2] / len;
        sg_eval_sh<D>(sh, dir, o.rgb, o.clampbits);
    }
}

// Stores the projected record and bins the Gaussian.  MUST be reached by EVERY thread (live or not) of a
// 256-thread workgroup (it contains workgroup barriers).
//  * the Gaussian-major slots of the workgroup's (tile,Gaussian) pairs are reserved with ONE atomic per
//    workgroup (prefix sums over tiles_touched); a wave's slots [base, base+total) are contiguous and in
//    lane order, which the backward relies on to stream a wave's gradient records coalesced;
//  * the pairs are then expanded load-balanced: lane l of pass c handles pair 64c + l of the wave
//    (binary search of the owner in the wave's prefix sums held in LDS), adds 1 to its tile's
//    counter with a RETURNING atomic and records (Gaussian, tile, arrival rank) -- the scatter that
//    follows the tile scan then needs no atomics at all.  With `hist` (T words of workgroup LDS, images of few
//    tiles) the ranks are first taken from an LDS histogram of the workgroup and rebased with one global atomic per
//    touched tile.
//  * DIRECT (sg_direct_keys: short lists vouched for, no histogram): the pair's key goes straight into row `tile` of bn.tile_keys
//    at that rank -- the (Gaussian, tile, rank) records, and the scatter pass that read them back, are gone; a rank beyond the row
//    is dropped (header[8] tells the scan, nothing is composited).  With `hist` (SG_FLAG_LONG_ROWS, rows of 16384) the key is stored
//    once the rank is final, and the lane clears the pair's rec_valid byte as the scatter pass used to.
// `scratch`: >= 256 words of LDS private to this wave.
// (the key store of direct binning as a non-temporal store: the preprocess 36 -> 48-57 us at cfg3; as an agent-scope write-through
// store (sc1): 37, and 1.5 % fewer views/s.  Plain stores.)
#define SG_KEY_STORE(p, v) (*(p) = (v))
__device__ __forceinline__ void sg_store_proj(bool live, int idx, const SgProj &o, float opac, SgGeom g, SgBin bn,
                                              int gx, uint32_t cap, int32_t *__restrict__ radii,
                                              uint32_t *__restrict__ scratch, uint32_t *__restrict__ hist, int T, uint32_t key_pitch)
{
    const bool direct = key_pitch != 0u;                  // (kernel-uniform) rows of key_pitch keys; with `hist`: SG_FLAG_LONG_ROWS
    uint32_t *sIncl = scratch, *sMin = scratch + 64, *sWid = scratch + 128, *sDep = scratch + 192;
    const int lane = threadIdx.x & 63;
    uint32_t incl = o.tt;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        uint32_t v = __shfl_up(incl, s, 64);
        if (lane >= s) incl += v;
    }
    const uint32_t total = __shfl(incl, 63, 64);
    // ONE allocator atomic per 256-thread workgroup: a single counter word sustains only ~88 returning
    // atomics per microsecond (MI355X_MICROARCH.md "dequeue"), one per wave would cost ~35 us at 2e5 Gaussians
    __shared__ uint32_t sWaveTot[4];
    __shared__ uint32_t sBlockBase;
    const int wave_ = (threadIdx.x >> 6) & 3;
    if (lane == 63) sWaveTot[wave_] = total;
    if (hist)                                     // few-tiles regime: this workgroup's per-tile pair counts live in LDS
        for (int t = threadIdx.x; t < T; t += blockDim.x) hist[t] = 0u;
    __syncthreads();
    // The allocator's answer is needed only where slots are WRITTEN (recC's offset, the pair records): every wave first issues its
    // returning tile-counter atomics -- the allocator word serves ~88 requests per microsecond chip-wide (782 workgroups: ~9 us of
    // queue), which used to be a barrier everybody sat at.
    const uint32_t rmin = (uint32_t)o.x0 | ((uint32_t)o.y0 << 16);
    const uint32_t rwh = (uint32_t)(o.x1 - o.x0) | ((uint32_t)(o.y1 - o.y0) << 16);
    if (live) {
        radii[idx] = o.mr;
        g.recA[SG_REC_STRIDE * (size_t)idx] = make_float4(o.pix[0], o.pix[1], o.conic[0], o.conic[1]);
        g.recB[SG_REC_STRIDE * (size_t)idx] = make_float4(o.conic[2], o.mr ? opac : 0.0f, o.rgb[0], o.rgb[1]);
        g.depth[idx] = o.depth;
        g.flags[idx] = o.clampbits;
    }
    const bool expand = total != 0 || hist;     // wave-uniform
    sIncl[lane] = incl; sMin[lane] = rmin; sWid[lane] = rwh & 0xffffu; sDep[lane] = __float_as_uint(o.depth);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    const int g0 = idx - lane;
    // four pairs per lane per round: the four returning atomics are in flight together
    uint32_t tile[4], local[4], gj[4], ctr[4], dep[4];
    auto take = [&](uint32_t p0) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t p = p0 + 64 * u + lane;
            tile[u] = 0; local[u] = 0; gj[u] = 0; ctr[u] = 0; dep[u] = 0;
            if (p < total) {
                int lo = 0, hi = 63;                  // smallest j with incl[j] > p
#pragma unroll
                for (int it = 0; it < 6; it++) {
                    int mid = (lo + hi) >> 1;
                    if (sIncl[mid] > p) hi = mid; else lo = mid + 1;
                }
                const int j = lo;
                const uint32_t excl = j ? sIncl[j - 1] : 0u;
                const uint32_t t = p - excl, w = sWid[j], mn = sMin[j];
                const uint32_t ty = (uint32_t)(((float)t + 0.5f) / (float)w);      // exact floor: t, w < 2^16
                const uint32_t tx = t - ty * w;
                tile[u] = ((mn >> 16) + ty) * (uint32_t)gx + (mn & 0xffffu) + tx;
                gj[u] = (uint32_t)(g0 + j);
                // rank inside this workgroup's share of the tile (LDS) / inside the tile (global): counters and histogram words
                // are indexed alike, in 4x4 blocks of tiles (sg_ctr_index)
                ctr[u] = sg_ctr_index((mn & 0xffffu) + tx, (mn >> 16) + ty, (uint32_t)gx);
                local[u] = hist ? atomicAdd(&hist[ctr[u]], 1u) : atomicAdd(&bn.tile_count[ctr[u]], 1u);
                dep[u] = sDep[j];
            }
        }
    };
    if (expand) take(0u);                              // the first (at cfg3: the only) round is in flight across the barrier
    // the allocator request goes out BEHIND this wave's tile-counter atomics (hipcc waits for a returning atomic right where it
    // is issued -- its wave-aggregation wrapper reads the result with v_readfirstlane -- so issued first it stalled wave 0 in front
    // of its own tile atomics: the workgroup's critical path was the allocator queue PLUS a round of tile atomics)
    if (threadIdx.x == 0) {
        uint32_t t = sWaveTot[0] + sWaveTot[1] + sWaveTot[2] + sWaveTot[3];
        sBlockBase = t ? atomicAdd(&bn.header[2], t) : 0u;
    }
    __syncthreads();
    uint32_t base = sBlockBase;
    for (int w = 0; w < wave_; w++) base += sWaveTot[w];
    if (live) {
        g.recC[SG_REC_STRIDE * (size_t)idx] = make_float4(o.rgb[2], __uint_as_float(base + incl - o.tt), __uint_as_float(rmin), __uint_as_float(rwh));
        g.slot[idx] = make_uint2(base + incl - o.tt, rwh);       // what the per-Gaussian backward needs of it, as a dense array
        // (writing the 16 B of padding as well, so that the whole line is dirty: no difference, 31.2 vs 31.3 us)
    }
    // round 0 (all of the wave's pairs at cfg3 / most of them on an avatar) stays in registers until its ranks are final: with the
    // LDS histogram they are rebased here instead of being written, re-read and rewritten
    uint32_t tile0[4], local0[4], gj0[4], ctr0[4], dep0[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { tile0[u] = tile[u]; local0[u] = local[u]; gj0[u] = gj[u]; ctr0[u] = ctr[u]; dep0[u] = dep[u]; }
    if (expand) {
        for (uint32_t p0 = 256; p0 < total; p0 += 256) {
            take(p0);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t p = p0 + 64 * u + lane;
                const uint32_t slot = base + p;
                if (direct && !hist) {
                    if (p < total) {
                        if (local[u] < key_pitch) SG_KEY_STORE(&bn.tile_keys[(size_t)tile[u] * key_pitch + local[u]], ((uint64_t)dep[u] << 32) | gj[u]);
                        else bn.header[8] = 1u;                 // a list longer than its row: sticky until the scan has seen it
                    }
                } else if (p < total && slot < cap) { bn.pair_gid[slot] = gj[u]; bn.pair_tile[slot] = tile[u]; bn.pair_local[slot] = local[u]; }
            }
        }
    }
    if (hist) {
        // one global returning atomic per (workgroup, touched tile) instead of one per pair -- on an avatar frame the
        // per-pair atomics were 55 of the 80 us of this kernel -- then the workgroup-local ranks are rebased.  Eight tiles per
        // thread and round, the atomics of a round all in flight before the first answer is used (one at a time they were T / 256
        // dependent round trips per thread).
        __syncthreads();
        for (int t0 = threadIdx.x; t0 < T; t0 += 8 * blockDim.x) {
            uint32_t cnt[8], ans[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int t = t0 + u * blockDim.x;
                cnt[u] = t < T ? hist[t] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) ans[u] = cnt[u] ? atomicAdd(&bn.tile_count[t0 + u * blockDim.x], cnt[u]) : 0u;   // (word = counter index)
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (cnt[u]) hist[t0 + u * blockDim.x] = ans[u];
        }
        __syncthreads();
        for (uint32_t p = 256 + lane; p < total; p += 64) {      // later rounds: a lane re-reads exactly the slots it wrote above
            const uint32_t slot = base + p;
            if (slot >= cap) continue;
            if (direct) {
                // (its records were the lane's notes: the key goes to the tile's row at the final rank; the depth of a Gaussian of this
                // workgroup was stored in front of the barriers above)
                const uint32_t gid = bn.pair_gid[slot], tl = bn.pair_tile[slot];
                const uint32_t rank = bn.pair_local[slot] + hist[sg_ctr_of_tile(tl, (uint32_t)gx)];
                if (rank < key_pitch) SG_KEY_STORE(&bn.tile_keys[(size_t)tl * key_pitch + rank], ((uint64_t)__float_as_uint(g.depth[gid]) << 32) | gid);
                else bn.header[8] = 1u;
                bn.rec_valid[slot] = 0;                          // (what the scatter pass did, one lane per pair: see SgBin::rec_valid)
            } else bn.pair_local[slot] += hist[sg_ctr_of_tile(bn.pair_tile[slot], (uint32_t)gx)];
        }
    }
    if (expand) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t p = 64 * u + lane;
            const uint32_t slot = base + p;
            if (direct) {
                if (p < total) {
                    const uint32_t rank = local0[u] + (hist ? hist[ctr0[u]] : 0u);
                    if (rank < key_pitch) SG_KEY_STORE(&bn.tile_keys[(size_t)tile0[u] * key_pitch + rank], ((uint64_t)dep0[u] << 32) | gj0[u]);
                    else bn.header[8] = 1u;
                    if (hist && slot < cap) bn.rec_valid[slot] = 0;
                }
            } else if (p < total && slot < cap) {
                bn.pair_gid[slot] = gj0[u]; bn.pair_tile[slot] = tile0[u];
                bn.pair_local[slot] = local0[u] + (hist ? hist[ctr0[u]] : 0u);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// ---- coalesced staging of 48-float rows ([P,16,3] SH coefficients / their gradients) through LDS:
// the wave's 64 rows are one contiguous 12-KiB block, moved with 16-B-per-lane accesses; LDS rows are
// 52 floats apart so that both the row-major fill and the one-row-per-lane ds_read_b128 are conflict-free.
#define SG_ROW_LDS 52
__device__ __forceinline__ void sg_rows48_load(const float *__restrict__ base, int g0, int P, int lane,
                                               float *__restrict__ l)
{
    const float4 *src = (const float4 *)(base + (size_t)g0 * 48);
    const int nf4 = (P - g0 < 64 ? P - g0 : 64) * 12;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const int f = i * 64 + lane;
        if (f < nf4) {
            const float4 v = src[f];
            const int row = f / 12, c4 = f - row * 12;
            *(float4 *)(l + row * SG_ROW_LDS + 4 * c4) = v;
        }
    }
}
// stores `rows` (<= 64) rows starting at Gaussian g0
// accumulate: dst += rows (the gradient buffer is shared by the views of a step: sg_rasterize_backward_gaussians)
__device__ __forceinline__ void sg_rows48_store(float *__restrict__ base, int g0, int P, int lane,
                                                const float *__restrict__ l, int rows = 64, bool accumulate = false)
{
    float4 *dst = (float4 *)(base + (size_t)g0 * 48);
    const int nf4 = (P - g0 < rows ? (P - g0 > 0 ? P - g0 : 0) : rows) * 12;
    if (accumulate) {
        float4 old[12];
#pragma unroll
        for (int i = 0; i < 12; i++) { const int f = i * 64 + lane; old[i] = f < nf4 ? dst[f] : make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const int f = i * 64 + lane;
            if (f < nf4) {
                const int row = f / 12, c4 = f - row * 12;
                const float4 v = *(const float4 *)(l + row * SG_ROW_LDS + 4 * c4);
                dst[f] = make_float4(old[i].x + v.x, old[i].y + v.y, old[i].z + v.z, old[i].w + v.w);
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const int f = i * 64 + lane;
        if (f < nf4) {
            const int row = f / 12, c4 = f - row * 12;
            dst[f] = *(const float4 *)(l + row * SG_ROW_LDS + 4 * c4);
        }
    }
}

struct SgGaussGrad {
    float dmean[3], g2[2], dop, dcol[3], dsc[3], drot[4], g6[6];
};

// Sum of this Gaussian's (tile,Gaussian) gradient records, fixed order -> deterministic.
__device__ __forceinline__ void sg_sum_records(SgRec grec, size_t cap, float4 recC, float a9[9])
{
    uint32_t goff = __float_as_uint(recC.y), wh = __float_as_uint(recC.w);
    uint32_t tt = (wh & 0xffffu) * (wh >> 16);
#pragma unroll
    for (int i = 0; i < 9; i++) a9[i] = 0.0f;
    for (uint32_t k = 0; k < tt; k++) {
        size_t r = (size_t)goff + k;
        if (r >= cap) break;
        float4 r0 = grec.a[2 * r], r1 = grec.a[2 * r + 1];
        a9[0] += r0.x; a9[1] += r0.y; a9[2] += r0.z; a9[3] += r0.w;
        a9[4] += r1.x; a9[5] += r1.y; a9[6] += r1.z; a9[7] += r1.w; a9[8] += grec.b[r];
    }
}

// Cooperative variant: the wave's records are ONE contiguous range (slots are reserved per wave in lane
// order), so the wave streams them with 16-B-per-lane loads into LDS (chunks of SG_REC_CHUNK records)
// and every lane then sums its own records from LDS in the same fixed order.  `l` >= SG_REC_CHUNK*12 floats.
#define SG_REC_CHUNK 128
// `valid` (few-tile frames; nullptr = every record was written): one byte per record, 1 = the sparse backward composite wrote it.
// Records whose byte is 0 are NOT loaded -- they enter the sums as the zeros the round-3 kernel stored there (still added: x + 0
// keeps the bits of every x the sums can hold, so the result is the one of the zero-filled buffer).  Lane L holds the bytes of
// records L and L + 64 of the chunk as two bits; the bits of the NEXT chunk are fetched beside the records of this one.
__device__ __forceinline__ uint32_t sg_valid_bits(const uint8_t *__restrict__ valid, uint32_t c0, uint32_t whi, int lane)
{
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const uint32_t r = c0 + (uint32_t)lane + 64u * i;
        if (r < whi) m |= (uint32_t)valid[r] << i;
    }
    return m;
}
__device__ __forceinline__ void sg_sum_records_coop(SgRec grec, size_t cap, bool vis, float4 recC,
                                                    int lane, float *__restrict__ l, float a9[9],
                                                    const uint8_t *__restrict__ valid = nullptr)
{
    const uint32_t goff = __float_as_uint(recC.y), wh = __float_as_uint(recC.w);
    const uint32_t tt = vis ? (wh & 0xffffu) * (wh >> 16) : 0u;
    const uint32_t lo = vis ? goff : 0xffffffffu, hi = vis ? goff + tt : 0u;
    uint32_t wlo = lo, whi = hi;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t a = __shfl_xor(wlo, o, 64), b = __shfl_xor(whi, o, 64);
        wlo = a < wlo ? a : wlo; whi = b > whi ? b : whi;
    }
#pragma unroll
    for (int i = 0; i < 9; i++) a9[i] = 0.0f;
    if (whi <= wlo) return;                               // wave-uniform: nothing visible
    if ((size_t)whi > cap) whi = (uint32_t)cap;
    static_assert(SG_REC_CHUNK == 128, "sg_valid_bits holds two records per lane");
    uint32_t vnext = valid ? sg_valid_bits(valid, wlo, whi, lane) : 3u;
    for (uint32_t c0 = wlo; c0 < whi; c0 += SG_REC_CHUNK) {
        const uint32_t n = whi - c0 < SG_REC_CHUNK ? whi - c0 : SG_REC_CHUNK;
        const float4 *src = grec.a + 2 * (size_t)c0;
        const float *srb = grec.b + (size_t)c0;
        if (valid) {                                      // (wave-uniform)
            const uint32_t vcur = vnext;
            if (c0 + SG_REC_CHUNK < whi) vnext = sg_valid_bits(valid, c0 + SG_REC_CHUNK, whi, lane);
            const float4 z4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            float4 v[SG_REC_CHUNK * 2 / 64];
            float w[SG_REC_CHUNK / 64];
            const uint32_t last = 2 * n - 1;
#pragma unroll
            for (int i = 0; i < SG_REC_CHUNK * 2 / 64; i++) {
                const uint32_t f0 = (uint32_t)lane + 64u * i, f = f0 < last ? f0 : last, r = f >> 1;
                const uint32_t ok = ((uint32_t)__shfl((int)vcur, (int)(r & 63u), 64) >> (r >> 6)) & 1u;
                v[i] = z4;
                if (ok) v[i] = src[f];
            }
#pragma unroll
            for (int i = 0; i < SG_REC_CHUNK / 64; i++) {
                const uint32_t f0 = (uint32_t)lane + 64u * i, f = f0 < n - 1 ? f0 : n - 1;
                const uint32_t ok = ((uint32_t)__shfl((int)vcur, (int)(f & 63u), 64) >> (f >> 6)) & 1u;
                w[i] = 0.0f;
                if (ok) w[i] = srb[f];
            }
#pragma unroll
            for (int i = 0; i < SG_REC_CHUNK * 2 / 64; i++) {
                const uint32_t f0 = (uint32_t)lane + 64u * i, f = f0 < last ? f0 : last;
                ((float4 *)l)[3 * (f >> 1) + (f & 1u)] = v[i];
            }
#pragma unroll
            for (int i = 0; i < SG_REC_CHUNK / 64; i++) {
                const uint32_t f0 = (uint32_t)lane + 64u * i, f = f0 < n - 1 ? f0 : n - 1;
                l[12 * f + 8] = w[i];
            }
        } else
        // all four 16-B loads and the two 4-B loads of the chunk are issued before the first one is used (as a loop the compiler
        // emitted load -> wait -> LDS write, ONE request in flight per wave: a memory latency per KiB); indices are clamped instead
        // of tested, so nothing branches around a load, and the surplus lanes rewrite the last element.  In LDS a record keeps
        // the 48-byte pitch of rounds 1-2 (two vectors + the ninth value in a third): at the 32-byte pitch of the memory layout
        // the lanes of a wave -- ~4 records apart: 128 B -- all hit the same banks (same-box A/B: 37.6 vs 33.5 us per launch).
        {
            float4 v[SG_REC_CHUNK * 2 / 64];
            float w[SG_REC_CHUNK / 64];
            const uint32_t last = 2 * n - 1;
#pragma unroll
            for (int i = 0; i < SG_REC_CHUNK * 2 / 64; i++) {
                const uint32_t f = (uint32_t)lane + 64u * i;
                v[i] = src[f < last ? f : last];
            }
#pragma unroll
            for (int i = 0; i < SG_REC_CHUNK / 64; i++) {
                const uint32_t f = (uint32_t)lane + 64u * i;
                w[i] = srb[f < n - 1 ? f : n - 1];
            }
#pragma unroll
            for (int i = 0; i < SG_REC_CHUNK * 2 / 64; i++) {
                const uint32_t f0 = (uint32_t)lane + 64u * i, f = f0 < last ? f0 : last;
                ((float4 *)l)[3 * (f >> 1) + (f & 1u)] = v[i];
            }
#pragma unroll
            for (int i = 0; i < SG_REC_CHUNK / 64; i++) {
                const uint32_t f0 = (uint32_t)lane + 64u * i, f = f0 < n - 1 ? f0 : n - 1;
                l[12 * f + 8] = w[i];
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        const uint32_t k0 = lo > c0 ? lo : c0, k1 = hi < c0 + n ? hi : c0 + n;
        for (uint32_t k = k0; k < k1; k++) {
            const float4 *r = (const float4 *)l + 3 * (k - c0);
            const float4 r0 = r[0], r1 = r[1];
            a9[0] += r0.x; a9[1] += r0.y; a9[2] += r0.z; a9[3] += r0.w;
            a9[4] += r1.x; a9[5] += r1.y; a9[6] += r1.z; a9[7] += r1.w; a9[8] += r[2].x;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Backward of sg_project_fwd for one VISIBLE Gaussian.  a9 = summed record
// (mean2D.x, mean2D.y, conic.x, conic.y, conic.w, opacity, colour r,g,b).
// dsh_out: (D+1)^2 x 3 gradient rows of this Gaussian (written iff want_sh; pass a local array); `first` false: ADDED to what the
// array holds (the frames of one step summed in registers).
template <int D>
__device__ __forceinline__ void sg_project_bwd(const SgCam &c, const float p[3], const float s3[3], const float q[4],
                                               const float *__restrict__ c6pre, const float *__restrict__ sh,
                                               uint32_t clampbits, const float a9[9], bool want_sh,
                                               float *__restrict__ dsh_out, SgGaussGrad &G, bool first = true)
{
    float *dmean = G.dmean, *g2 = G.g2, *dcol = G.dcol, *dsc = G.dsc, *drot = G.drot, *g6 = G.g6;
    g2[0] = a9[0]; g2[1] = a9[1];
    float dLx = a9[2], dLy = a9[3], dLz = a9[4];
    G.dop = a9[5]; dcol[0] = a9[6]; dcol[1] = a9[7]; dcol[2] = a9[8];
#pragma unroll
    for (int k = 0; k < 3; k++) dsc[k] = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; k++) drot[k] = 0.0f;
#pragma unroll
    for (int k = 0; k < 6; k++) g6[k] = 0.0f;
    float c6[6];
    if (c6pre) {
#pragma unroll
        for (int k = 0; k < 6; k++) c6[k] = c6pre[k];
    } else {
        sg_cov3d(s3, c.mod, q, c6);
    }
    // ---- cov2D backward
    float pv[3], Mm[6], tc[3], abc[3]; bool xin, yin;
    sg_xf4x3(p, c.view, pv);
    sg_proj_jac(pv, c.fx, c.fy, c.tanfovx, c.tanfovy, c.view, Mm, tc, xin, yin);
    sg_cov2d(Mm, c6, abc);
    float a = abc[0], b = abc[1], cc = abc[2];
    float denom = a * cc - b * b;
    float denom2inv = 1.0f / (denom * denom + 0.0000001f);
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    if (denom2inv != 0.0f) {
        dL_da = denom2inv * (-cc * cc * dLx + 2.0f * b * cc * dLy + (denom - a * cc) * dLz);
        dL_dc = denom2inv * (-a * a * dLz + 2.0f * a * b * dLy + (denom - a * cc) * dLx);
        dL_db = denom2inv * 2.0f * (b * cc * dLx - (denom + 2.0f * b * b) * dLy + a * b * dLz);
        const float *m0 = Mm, *m1 = Mm + 3;
        g6[0] = m0[0] * m0[0] * dL_da + m0[0] * m1[0] * dL_db + m1[0] * m1[0] * dL_dc;
        g6[3] = m0[1] * m0[1] * dL_da + m0[1] * m1[1] * dL_db + m1[1] * m1[1] * dL_dc;
        g6[5] = m0[2] * m0[2] * dL_da + m0[2] * m1[2] * dL_db + m1[2] * m1[2] * dL_dc;
        g6[1] = 2.0f * m0[0] * m0[1] * dL_da + (m0[0] * m1[1] + m0[1] * m1[0]) * dL_db + 2.0f * m1[0] * m1[1] * dL_dc;
        g6[2] = 2.0f * m0[0] * m0[2] * dL_da + (m0[0] * m1[2] + m0[2] * m1[0]) * dL_db + 2.0f * m1[0] * m1[2] * dL_dc;
        g6[4] = 2.0f * m0[2] * m0[1] * dL_da + (m0[1] * m1[2] + m0[2] * m1[1]) * dL_db + 2.0f * m1[1] * m1[2] * dL_dc;
    }
    float gM[6];
    {
        float V[9] = { c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5] };
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float v0 = Mm[0] * V[3 * k] + Mm[1] * V[3 * k + 1] + Mm[2] * V[3 * k + 2];
            float v1 = Mm[3] * V[3 * k] + Mm[4] * V[3 * k + 1] + Mm[5] * V[3 * k + 2];
            gM[k] = 2.0f * v0 * dL_da + v1 * dL_db;
            gM[3 + k] = 2.0f * v1 * dL_dc + v0 * dL_db;
        }
    }
    const float *view = c.view, *proj = c.proj;
    float dJ00 = gM[0] * view[0] + gM[1] * view[4] + gM[2] * view[8];
    float dJ02 = gM[0] * view[2] + gM[1] * view[6] + gM[2] * view[10];
    float dJ11 = gM[3] * view[1] + gM[4] * view[5] + gM[5] * view[9];
    float dJ12 = gM[3] * view[2] + gM[4] * view[6] + gM[5] * view[10];
    float tz = 1.0f / tc[2], tz2 = tz * tz, tz3 = tz2 * tz;
    float dtx = (xin ? 1.0f : 0.0f) * -c.fx * tz2 * dJ02;
    float dty = (yin ? 1.0f : 0.0f) * -c.fy * tz2 * dJ12;
    float dtz = -c.fx * tz2 * dJ00 - c.fy * tz2 * dJ11 + (2.0f * c.fx * tc[0]) * tz3 * dJ02 + (2.0f * c.fy * tc[1]) * tz3 * dJ12;
    dmean[0] = view[0] * dtx + view[1] * dty + view[2] * dtz;
    dmean[1] = view[4] * dtx + view[5] * dty + view[6] * dtz;
    dmean[2] = view[8] * dtx + view[9] * dty + view[10] * dtz;
    // ---- projection backward
    float mh[4];
    sg_xf4x4(p, proj, mh);
    float mw = 1.0f / (mh[3] + 0.0000001f);
    float mul1 = (proj[0] * p[0] + proj[4] * p[1] + proj[8] * p[2] + proj[12]) * mw * mw;
    float mul2 = (proj[1] * p[0] + proj[5] * p[1] + proj[9] * p[2] + proj[13]) * mw * mw;
    dmean[0] += (proj[0] * mw - proj[3] * mul1) * g2[0] + (proj[1] * mw - proj[3] * mul2) * g2[1];
    dmean[1] += (proj[4] * mw - proj[7] * mul1) * g2[0] + (proj[5] * mw - proj[7] * mul2) * g2[1];
    dmean[2] += (proj[8] * mw - proj[11] * mul1) * g2[0] + (proj[9] * mw - proj[11] * mul2) * g2[1];
    // ---- cov3D backward
    if (!c6pre) {
        float Gs[9] = { g6[0], 0.5f * g6[1], 0.5f * g6[2], 0.5f * g6[1], g6[3], 0.5f * g6[4],
                        0.5f * g6[2], 0.5f * g6[4], g6[5] };
        float R[9];
        sg_quat_to_R(q, R);
        float s[3] = { c.mod * s3[0], c.mod * s3[1], c.mod * s3[2] };
        float GR[9], dR[9];
#pragma unroll
        for (int a2 = 0; a2 < 3; a2++)
#pragma unroll
            for (int k = 0; k < 3; k++)
                GR[3 * a2 + k] = Gs[3 * a2] * R[k] + Gs[3 * a2 + 1] * R[3 + k] + Gs[3 * a2 + 2] * R[6 + k];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float rgr = R[k] * GR[k] + R[3 + k] * GR[3 + k] + R[6 + k] * GR[6 + k];
            dsc[k] = 2.0f * s[k] * rgr;       // w.r.t. the MODIFIED scale, as upstream reports it (no factor mod)
#pragma unroll
            for (int a2 = 0; a2 < 3; a2++) dR[3 * a2 + k] = 2.0f * GR[3 * a2 + k] * s[k] * s[k];
        }
        float r = q[0], x = q[1], y = q[2], z = q[3];
        drot[0] = 2.0f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
        drot[1] = 2.0f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.0f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2.0f * x * dR[8]);
        drot[2] = 2.0f * (-2.0f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2.0f * y * dR[8]);
        drot[3] = 2.0f * (-2.0f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.0f * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
    }
    // ---- SH backward (and its contribution to dL/dmean through the view direction)
    if (want_sh) {
        constexpr int nc = (D + 1) * (D + 1);
        float dorig[3] = { p[0] - c.campos[0], p[1] - c.campos[1], p[2] - c.campos[2] };
        float sum2 = dorig[0] * dorig[0] + dorig[1] * dorig[1] + dorig[2] * dorig[2];
        float len = sqrtf(sum2);
        float dir[3] = { dorig[0] / len, dorig[1] / len, dorig[2] / len };
        float bas[16], db[48];
        sg_sh_basis<D>(dir, bas);
        sg_sh_basis_grad<D>(dir, db);
        float dRGB[3];
#pragma unroll
        for (int ch = 0; ch < 3; ch++) dRGB[ch] = (clampbits >> ch) & 1u ? 0.0f : dcol[ch];
        float ddir[3] = { 0, 0, 0 };
#pragma unroll
        for (int k = 0; k < nc; k++)
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                // frames of one step (sg_preprocess_bwd_kernel / sg_skin_bwd_kernel): frame 0 assigns, later frames add
                dsh_out[3 * k + ch] = first ? bas[k] * dRGB[ch] : dsh_out[3 * k + ch] + bas[k] * dRGB[ch];
                if (k > 0) {
                    float sv = sh[3 * k + ch] * dRGB[ch];
                    ddir[0] += db[3 * k] * sv; ddir[1] += db[3 * k + 1] * sv; ddir[2] += db[3 * k + 2] * sv;
                }
            }
        float inv32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
        float vx = dorig[0], vy = dorig[1], vz = dorig[2];
        dmean[0] += ((sum2 - vx * vx) * ddir[0] - vy * vx * ddir[1] - vz * vx * ddir[2]) * inv32;
        dmean[1] += (-vx * vy * ddir[0] + (sum2 - vy * vy) * ddir[1] - vz * vy * ddir[2]) * inv32;
        dmean[2] += (-vx * vz * ddir[0] - vy * vz * ddir[1] + (sum2 - vz * vz) * ddir[2]) * inv32;
    }
}
