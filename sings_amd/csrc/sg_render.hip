// Per-tile alpha-composite forward / backward.
//
// Replaces renderCUDA<3> forward and backward of the rasterizer the reference calls at
// sings/rec/renderer/gs_renderer_single.py:87-95.
//
// Both kernels are VALU-issue bound (not HBM): ~2e8 pixel x Gaussian evaluations per direction at
// cfg3.  wave64 design:
//  * One 256-thread workgroup per 16x16 tile; wave w owns the 8x8 pixel QUADRANT w (lane = pixel).
//  * The tile's depth-sorted list is staged cooperatively through LDS (three aligned 16-B gathers out of ONE 64-B record per
//    entry, one entry per thread).  While staging, each thread computes for ITS entry which quadrants the
//    alpha >= 1/255 ellipse can reach (exact ellipse-vs-rectangle test with safety margins); every wave
//    then compacts (ballot) the entries that reach its quadrant and loops over those only: ~2.4x fewer
//    wave-level evaluations than the upstream 16x16 block, identical results, and 4 independent waves
//    per tile keep avatar close-ups (few, very long tile lists) busy.
//    The rectangle is the bounding box of the quadrant's LIVE pixels (unsaturated ones), so entries hidden behind
//    saturated pixels drop out; the forward stores the mask byte per sorted entry and the backward reuses it.
//  * The per-entry body is straight-line predicated code (no exec-mask branches).
//  * Lists longer than 256 entries: the forward checkpoints (T, colour) every 256 entries and the backward runs one
//    workgroup per (tile, 256-entry depth segment) instead of one serial walk per tile.
//  * Backward: the 9 per-pixel partials of an entry are summed across the wave's 64 lanes with a
//    multi-value butterfly (v_permlane32_swap / v_permlane16_swap / DPP: 24 ops for 9 values instead of
//    54), the <= 4 quadrant sums are combined in LDS in a fixed order and stored ONCE per (tile,Gaussian)
//    as a 36-byte record at the Gaussian-major slot reserved in the forward pass.  No float atomics
//    (memory-side atomics cap at ~1.3 TB/s on MI355X and scattered single-row adds are 17x slower);
//    gradients are bitwise reproducible.
#include "sg_sort.h"

#define SG_FB 256         // forward: list entries staged per batch (one per thread)
#define SG_BB 64         // backward: entries per batch (bounded by the LDS of the quadrant-sum buffer)
#define SG_UNSET 0xffffffffu   // bit pattern (a NaN) of a quadrant-sum slot nobody wrote

__device__ __forceinline__ float sg_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
// The composite loops evaluate alpha = min(.99, o 2^p) with p = log2(e) * power: the staging thread scales the conic ONCE per
// (tile, entry) -- (A', B', C') = (-log2(e)/2 A, -log2(e) B, -log2(e)/2 C) -- and the pixel loop needs five operations,
// p = (A' dx + B' dy) dx + C' dy dy, instead of eight plus the multiplication by log2(e).  Forward and backward use the SAME
// expression, so they take the same alpha >= 1/255 decisions.
#define SG_KA (-0.5f * 1.44269504088896340736f)
#define SG_KB (-1.44269504088896340736f)
__device__ __forceinline__ float sg_power2(float Ap, float Bp, float Cp, float dx, float dy)
{
    return fmaf(Cp * dy, dy, fmaf(Ap, dx, Bp * dy) * dx);
}

// XCD-aware map: workgroup b runs on XCD b % 8 (round-robin dispatch).  Each XCD takes runs of SG_XCD_RUN consecutive
// tiles (neighbouring tiles share Gaussians -> same L2), and the runs are dealt round-robin to the XCDs so that
// every XCD sees every part of the image: with one contiguous band of tiles per XCD an avatar (all the work in the
// middle rows) kept two of the eight XCDs almost idle.
#define SG_XCD_RUN 16
__device__ __forceinline__ int sg_tile_of_block(int block)
{
    const int xcd = block & 7, slot = block >> 3;
    // Workgroups of one XCD go round its 32 CUs (measured with s_getreg HW_ID: a CU receives blocks b, b + 256, ...), so
    // without the skew a CU always gets the same position of the run -- the same image column when the image is 32
    // tiles wide -- and the CUs that own an avatar's body columns carry 2.6x the mean load.  (slot / 32) * 7 walks the
    // position through the run from one visit of a CU to the next (max/mean load 1.6).  The avatar forward itself is
    // bound by its longest tile (39 serial batches), not by this; see LAB.md.
    const int within = (slot + 7 * (slot >> 5)) & (SG_XCD_RUN - 1);
    return ((slot / SG_XCD_RUN) * 8 + xcd) * SG_XCD_RUN + within;
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

// Bounding box (tile-relative pixel coordinates) of the lanes of an 8x8 quadrant whose bit is set in `live`
// (lane = 8 * row + column); an empty set gives an empty box (x0 > x1).  Wave-uniform scalar arithmetic.
__device__ __forceinline__ float4 sg_live_box(unsigned long long live, int wave)
{
    if (live == 0ull) return make_float4(1.0f, 0.0f, 1.0f, 0.0f);
    unsigned long long m = live;
    m |= m >> 32; m |= m >> 16; m |= m >> 8;
    const uint32_t cols = (uint32_t)m & 0xffu;
    unsigned long long t = live;
    t |= t >> 4; t |= t >> 2; t |= t >> 1;
    const uint32_t rows = (uint32_t)(((t & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56);
    const int c0 = __builtin_ctz(cols), c1 = 31 - __builtin_clz(cols), r0 = __builtin_ctz(rows), r1 = 31 - __builtin_clz(rows);
    const int qx = 8 * (wave & 1), qy = 8 * (wave >> 1);
    return make_float4((float)(qx + c0), (float)(qx + c1), (float)(qy + r0), (float)(qy + r1));
}

// The same for a 4x4 block (lanes 0..15, lane = 4 * row + column) whose tile-relative origin is (ox, oy): the waves of a
// workgroup that composites ONE quadrant of a long tile (sg_render_fwd_body, `split`).
__device__ __forceinline__ float4 sg_live_box16(unsigned long long live, int ox, int oy)
{
    const uint32_t m = (uint32_t)live & 0xffffu;
    if (m == 0u) return make_float4(1.0f, 0.0f, 1.0f, 0.0f);
    const uint32_t cols = (m | (m >> 4) | (m >> 8) | (m >> 12)) & 0xfu;
    const uint32_t rows = ((m & 0xfu) ? 1u : 0u) | ((m & 0xf0u) ? 2u : 0u) | ((m & 0xf00u) ? 4u : 0u) | ((m & 0xf000u) ? 8u : 0u);
    const int c0 = __builtin_ctz(cols), c1 = 31 - __builtin_clz(cols), r0 = __builtin_ctz(rows), r1 = 31 - __builtin_clz(rows);
    return make_float4((float)(ox + c0), (float)(ox + c1), (float)(oy + r0), (float)(oy + r1));
}

// box[q] = (x0, x1, y0, y1): the part of quadrant q that still matters (sg_live_box) -- the whole quadrant at first;
// once pixels saturate (forward) / for the pixels an entry can still have contributed to (backward) only the
// bounding box of the live pixels.  Entries that cannot reach it are dropped from that quadrant's list: this is what
// keeps silhouette tiles of an opaque body cheap (thousands of entries hidden behind saturated pixels, a few
// background pixels that never saturate).
__device__ __forceinline__ uint32_t sg_quad_mask(float4 a, float4 b, float X0, float Y0, const float4 *__restrict__ box)
{
    const float o255 = 255.0f * b.y;
    if (!(o255 >= 0.999f)) return 0u;
    const float bound = 2.0f * (__logf(o255) * 1.002f + 0.004f) + 0.01f;
    const float A = a.z, B = a.w, C = b.x;
    const float det = A * C - B * B;
    const bool illc = !(det > 1e-3f * A * C) || !(A > 0.0f) || !(C > 0.0f);     // ill-conditioned: keep everything live
    const float mx = a.x - X0, my = a.y - Y0;
    uint32_t m = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float4 bx = box[q];
        if (!(bx.x <= bx.y)) continue;
        float mq = illc ? 0.0f : sg_min_q_rect(A, B, C, mx, my, bx.x, bx.y, bx.z, bx.w);
        if (mq * 0.999f <= bound) m |= 1u << q;
    }
    return m;
}


// entries of the staged batch whose mask has bit `w`, compacted in list order (wave-local); the list holds index * MUL
template <int MUL>
__device__ __forceinline__ int sg_compact_quadrant(const uint32_t *__restrict__ sM, int cnt, int w, int lane,
                                                   unsigned long long lt, uint16_t *__restrict__ list, int batch)
{
    int nl = 0;
    for (int c = 0; c < batch; c += 64) {
        const int idx = c + lane;
        const bool bit = idx < cnt && ((sM[idx] >> w) & 1u);
        const unsigned long long bal = __ballot(bit);
        if (bit) list[nl + __popcll(bal & lt)] = (uint16_t)(idx * MUL);     // MUL: the caller's record pitch in bytes, or 1
        nl += __popcll(bal);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    return nl;
}

// PIPE: software-pipelined walk of a quadrant's list (see the loop); the plain loop otherwise.
#define SG_FWD_PARAMS int W, int H, int gx, int T, int nblocks, const uint2 *__restrict__ ranges,                          \
                      const uint64_t *__restrict__ pair_keys, uint32_t *__restrict__ point_list,                         \
                      uint64_t *__restrict__ point_keys, const float4 *__restrict__ recA,                                \
                      const float4 *__restrict__ recB, const float4 *__restrict__ recC,                                  \
                      const float *__restrict__ bg, float *__restrict__ out_color,                                       \
                      float *__restrict__ final_T, uint32_t *__restrict__ n_contrib,                                     \
                      const uint32_t *__restrict__ ck_start, float4 *__restrict__ ckpt, uint32_t ck_cap,                 \
                      uint32_t *__restrict__ header, uint8_t *__restrict__ pair_mask, uint32_t *__restrict__ tile_count,            \
                      const uint4 *__restrict__ long_items, uint32_t mask_plane, const uint4 *__restrict__ plan,                  \
                      uint32_t *__restrict__ item_w, uint32_t w_plane, uint32_t key_pitch
#define SG_FWD_ARGS W, H, gx, T, nblocks, ranges, pair_keys, point_list, point_keys, recA, recB, recC, bg, out_color, final_T, \
                    n_contrib, ck_start, ckpt, ck_cap, header, pair_mask, tile_count, long_items, mask_plane, plan, item_w, w_plane, key_pitch
// WEIGHTS: leave the backward's work-item weights (few-tile frames: item_w, first_item); always with PIPE
template <bool PIPE, bool WEIGHTS>
__device__ __forceinline__ void sg_render_fwd_body(SG_FWD_PARAMS)
{
    __shared__ float4 sR[SG_FB][3];            // staged entry: (mean x, mean y, A', B') (C', opacity, colour 0, 1) (colour 2, -, -, -)
    __shared__ uint32_t sM[SG_FB];
    __shared__ uint16_t sList[4][SG_FB];       // byte offsets into sR (index * 48)
    __shared__ float4 sBox[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // The counters the NEXT forward's preprocess counts into are consumed by now (the scan ran before this kernel): leave
    // them zeroed, so that a caller who keeps its workspace can skip the zeroing launch (SG_FLAG_WS_CLEAN).
    if (tid == 0 && blockIdx.x == 0) header[2] = 0u;
    // Few-tile frames (PIPE): a tile whose list is longer than 1024 entries -- an avatar's hands: thousands of small splats, no
    // pixel saturates before entry ~3000 -- is composited by FOUR workgroups, one per quadrant, each wave owning a 4x4 block
    // (16 active lanes).  A wave walks only the entries that reach ITS pixels, and it issues at most one instruction every ~5
    // cycles however idle the chip is (per-tile clocks: the deepest tile alone set the kernel time, 3000 passes x 46 ns), so
    // shorter per-wave lists on otherwise idle SIMDs are what shortens the kernel; results are bit-identical (a pixel sees the
    // same entries in the same order).  The first `nblocks` blocks are quadrants 1..3 of the long tiles (they are the heaviest
    // work: dispatched first), listed by the scan in `long_items`; the tile's own block takes quadrant 0.
    int tile, part = 0;
    bool split = false;
    if (PIPE && (int)blockIdx.x < nblocks) {
        const uint32_t li = blockIdx.x / 3u;
        if (header[1] || li >= header[4]) return;
        tile = (int)long_items[li].x; part = 1 + (int)(blockIdx.x % 3u); split = true;
    } else {
        tile = sg_tile_of_block((int)blockIdx.x - (PIPE ? nblocks : 0));
        if (tile >= T) return;
        if (tid == 0) tile_count[sg_ctr_of_tile((uint32_t)tile, (uint32_t)gx)] = 0u;
    }
    const int X0 = (tile % gx) * 16, Y0 = (tile / gx) * 16;
    const uint2 range = ranges[tile];
    // R > capacity: part of the sorted list was never written (the caller re-runs with a larger workspace) --
    // render the background instead of gathering through stale ids
    const int n = header[1] ? 0 : (int)(range.y - range.x);
    if (PIPE && nblocks > 0 && n > SG_WSORT_MAX) split = true;
    // pixel of this lane: quadrant `wave` of the tile, or (split) block `wave` of quadrant `part`
    const int bx0 = split ? 8 * (part & 1) + 4 * (wave & 1) : 8 * (wave & 1), by0 = split ? 8 * (part >> 1) + 4 * (wave >> 1) : 8 * (wave >> 1);
    const int lx = split ? (lane & 3) : (lane & 7), ly = split ? ((lane >> 2) & 3) : (lane >> 3);
    const int px = X0 + bx0 + lx, py = Y0 + by0 + ly;
    const bool inside = px < W && py < H && (!split || lane < 16);
    const float pxf = (float)px, pyf = (float)py;
    // slot of this pixel in a checkpoint (the backward's layout: quadrant * 64 + 8 * row + column inside the quadrant)
    const int ckidx = split ? part * 64 + 8 * (by0 + ly - 8 * (part >> 1)) + (bx0 + lx - 8 * (part & 1)) : tid;
    const bool ckok = !split || lane < 16;
    // quadrants an entry was composited in, for the backward pass: one byte per list entry; a split tile's four workgroups write
    // one plane each (bit `part`)
    uint8_t *__restrict__ pmask = pair_mask + (size_t)(split ? part : 0) * mask_plane;
    const uint32_t cks = n > SG_SEG ? ck_start[tile] : 0xffffffffu;   // segmented list: checkpoint slots cks + segment
    const uint32_t first_item = (PIPE || WEIGHTS) ? plan[tile].x : 0u;   // first backward work item of this tile
    float Tr = 1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
    uint32_t last = 0;
    bool done = !inside;
    const unsigned long long lt = (1ull << lane) - 1ull;
    // software pipeline over batches: the gathers of batch k+1 are in flight while batch k is composited (a tile
    // whose list is thousands of entries long runs alone on its CU -- nothing else hides the two dependent loads)
    float4 pa = make_float4(0, 0, 0, 0), pb = pa;
    float pc = 0.0f;
    static_assert((SG_WSORT_MAX + SG_RANKSORT_MAX) * 8 <= SG_FB * 48, "sort buffers alias sR");
    if (n > 0 && n <= SG_WSORT_MAX) {
        // Short list (<= 1024 entries: every list at cfg3, most of an avatar's): this workgroup sorts the tile's keys itself, in LDS, and hands the order to
        // the backward pass through point_list -- the separate sort pass of round 1 (18 us, all latency) is gone and the
        // sort of one tile overlaps the compositing of the other tiles resident on the CU.
        uint64_t *sKey = (uint64_t *)sR;                    // aliases the staging buffer: consumed before the first batch
        // (direct binning, key_pitch != 0: `pair_keys` is SgBin::tile_keys, the tile's unsorted keys are its row)
        sg_sort_short_list(key_pitch ? pair_keys + (size_t)tile * key_pitch : pair_keys + range.x, n, sKey, sKey + SG_WSORT_MAX, tid);
        for (int i = tid; i < n; i += SG_FB) {              // (later batches read their ids back from point_list: the barrier at
            const uint64_t key = sKey[i];                   //  the top of the batch loop orders these stores in front of those loads)
            const uint32_t gid = (uint32_t)key;
            point_list[range.x + i] = gid;
            if (point_keys) point_keys[range.x + i] = ((uint64_t)tile << 32) | (key >> 32);
            if (i == tid) { pa = recA[SG_REC_STRIDE * (size_t)gid]; pb = recB[SG_REC_STRIDE * (size_t)gid]; pc = recC[SG_REC_STRIDE * (size_t)gid].x; }
        }
    } else if (tid < n) {                                   // long list: sorted by sg_tile_sort_kernel / sg_tile_rank_kernel
        const uint32_t gid = point_list[range.x + tid];
        pa = recA[SG_REC_STRIDE * (size_t)gid]; pb = recB[SG_REC_STRIDE * (size_t)gid]; pc = recC[SG_REC_STRIDE * (size_t)gid].x;
    }
    for (int base = 0; base < n; base += SG_FB) {
        {   // this quadrant's pixels that are still being composited
            const float4 bx = split ? sg_live_box16(__ballot(!done), bx0, by0) : sg_live_box(__ballot(!done), wave);
            if (lane == 0) sBox[wave] = bx;
        }
        if (__syncthreads_count(done) == 256) break;       // also: the previous batch is fully consumed
        const int e = base + tid;
        bool staged = false;
        if (e < n) {
            sR[tid][0] = make_float4(pa.x, pa.y, SG_KA * pa.z, SG_KB * pa.w);
            sR[tid][1] = make_float4(SG_KA * pb.x, pb.y, pb.z, pb.w);
            sR[tid][2].x = pc;
            const uint32_t mk = sg_quad_mask(pa, pb, (float)X0, (float)Y0, sBox);
            sM[tid] = mk;
            staged = mk != 0u;
            pmask[range.x + e] = (uint8_t)(split ? (mk ? 1u << part : 0u) : mk);     // the backward composites exactly these (entry, quadrant) pairs
        }
        if (e + SG_FB < n) {
            const uint32_t gid = point_list[range.x + e + SG_FB];
            pa = recA[SG_REC_STRIDE * (size_t)gid]; pb = recB[SG_REC_STRIDE * (size_t)gid]; pc = recC[SG_REC_STRIDE * (size_t)gid].x;
        }
        if (PIPE || WEIGHTS) {
            // few-tile frames: the weight of the backward work item (tile, this segment) = entries composited somewhere; the
            // backward starts its heaviest items first (sg_order_items_kernel sorts them)
            const int wcount = __syncthreads_count(staged);
            const uint32_t wi = first_item + (uint32_t)(base / SG_SEG);
            if (tid == 0 && wi < w_plane) {
                if (split) atomicMax(&item_w[wi], (uint32_t)wcount);    // four workgroups, one item: its latency is the slowest quadrant's
                else item_w[wi] = (uint32_t)wcount;
            }
        } else
        __syncthreads();
        if (__ballot(done) == ~0ull) continue;             // this quadrant is finished (wave-uniform)
        // state in front of entry `base`, for the backward pass of the segments behind it (SG_FB == SG_SEG)
        if (base > 0 && ckok && cks != 0xffffffffu && cks + (uint32_t)(base / SG_SEG) < ck_cap)
            ckpt[(size_t)(cks + (uint32_t)(base / SG_SEG)) * 256 + ckidx] = make_float4(Tr, C0, C1, C2);
        const int cnt = n - base < SG_FB ? n - base : SG_FB;
        uint16_t *list = sList[wave];
        const int nl = sg_compact_quadrant<48>(sM, cnt, wave, lane, lt, list, SG_FB);
        uint32_t lastk = 0xffffffffu;                      // record offset of the last entry blended in this batch
        // one (entry, quadrant) pass -- straight-line, predicated: no exec-mask branches.  (A macro, not a lambda: captured by
        // reference, `done` lived in a byte register and every pass paid three extra vector instructions for it.)
#define SG_FWD_PASS(ga, gb, gc, ko)                                                                        \
        do {                                                                                               \
            const float dx = (ga).x - pxf, dy = (ga).y - pyf;                                              \
            const float power = sg_power2((ga).z, (ga).w, (gb).x, dx, dy);                                 \
            const float alpha = fminf(0.99f, (gb).y * __builtin_amdgcn_exp2f(power));                      \
            const float test_T = Tr * (1.0f - alpha);                                                      \
            const bool valid = !done & !(power > 0.0f) & !(alpha < 1.0f / 255.0f);                         \
            const bool term = valid & (test_T < 0.0001f);                                                  \
            const bool blend = valid & !term;                                                              \
            const float w = blend ? alpha * Tr : 0.0f;                                                     \
            C0 = fmaf((gb).z, w, C0); C1 = fmaf((gb).w, w, C1); C2 = fmaf((gc), w, C2);                    \
            Tr = blend ? test_T : Tr;                                                                      \
            lastk = blend ? (ko) : lastk;                                                                  \
            done = done | term;                                                                            \
        } while (0)
        // The same pass for the pipelined walk, as a lambda -- deliberately: captured by reference, `done` lives in a VECTOR register
        // there and the validity logic becomes vector instructions, where the macro form chains v_cmp -> s_and -> v_cndmask through
        // VCC / scalar registers.  With one wave per SIMD (the regime this walk exists for) every vector -> scalar -> vector hop
        // is exposed latency: the all-vector form is 10 us faster on the avatar frame (106 vs 116 us); with eight busy waves the
        // extra vector instructions cost 4 us at cfg3 (74.7 vs 70.7), hence the macro for the plain loop.  Same-box A/B both ways.
        auto pass = [&](const float4 ga, const float4 gb, const float gc, const uint32_t ko) { SG_FWD_PASS(ga, gb, gc, ko); };
        if (PIPE) {
            // Software pipeline over the quadrant's list: the record of entry i + 1 and the list word of entry i + 2 are
            // requested BEFORE the arithmetic of entry i (two register sets, A and B, alternate: no copies).  As a plain loop
            // every pass is list word -> wait -> record -> wait -> 19 VALU: two dependent LDS latencies exposed per pass.  With
            // eight busy waves on the SIMD other waves fill those stalls; the deepest tiles of an avatar frame run ALONE at the
            // end of the kernel (per-tile clocks, profiles/r03_fwd_experiments.log: 12 batches in 128 us, one wave per SIMD) and
            // there the stalls were more than half of every pass.
            auto rec_of = [&](uint32_t k) { return (const float4 *)((const char *)&sR[0][0] + k); };
            uint32_t kA = nl > 0 ? list[0] : 0u, kB = nl > 1 ? list[1] : kA;
            float4 aA = rec_of(kA)[0], bA = rec_of(kA)[1];
            float cA = rec_of(kA)[2].x;
            for (int i = 0; i < nl; i += 2) {
                const float4 aB = rec_of(kB)[0], bB = rec_of(kB)[1];
                const float cB = rec_of(kB)[2].x;
                const uint32_t kA2 = list[i + 2 < nl ? i + 2 : nl - 1];
                __builtin_amdgcn_sched_barrier(0);          // (hipcc otherwise sinks the requests to just in front of their use)
                pass(aA, bA, cA, kA);
                if (i + 1 >= nl) break;
                aA = rec_of(kA2)[0]; bA = rec_of(kA2)[1]; cA = rec_of(kA2)[2].x;
                const uint32_t kB2 = list[i + 3 < nl ? i + 3 : nl - 1];
                __builtin_amdgcn_sched_barrier(0);
                pass(aB, bB, cB, kB);
                kA = kA2; kB = kB2;
            }
        } else {
            for (int i = 0; i < nl; i++) {
                const uint32_t ko = list[i];
                const float4 *rec = (const float4 *)((const char *)&sR[0][0] + ko);
                const float4 ga = rec[0], gb = rec[1];
                const float gc = rec[2].x;
                SG_FWD_PASS(ga, gb, gc, ko);
            }
        }
#undef SG_FWD_PASS
        if (lastk != 0xffffffffu) last = (uint32_t)base + lastk / 48u + 1u;
    }
    if (cks < ck_cap && ckok) ckpt[(size_t)cks * 256 + ckidx] = make_float4(Tr, C0, C1, C2);   // slot 0: final state
    if (inside) {
        const size_t pid = (size_t)py * W + px, hw = (size_t)H * W;
        final_T[pid] = Tr;
        n_contrib[pid] = last;
        out_color[pid] = fmaf(Tr, bg[0], C0);
        out_color[hw + pid] = fmaf(Tr, bg[1], C1);
        out_color[2 * hw + pid] = fmaf(Tr, bg[2], C2);
    }
}

// The plain loop at eight waves per SIMD: frames of many tiles (cfg3: 8 160 tiles, lists of ~100 entries, every SIMD busy).
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
sg_render_fwd_kernel(SG_FWD_PARAMS) { sg_render_fwd_body<false, false>(SG_FWD_ARGS); }
// The pipelined loop: ONE frame of few tiles with lists of thousands of entries (an avatar in front of a background: T <= 4096, the
// regime of the LDS histogram in the preprocess), where the kernel ends in a handful of deep tiles, one wave per SIMD.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 7)))
sg_render_fwd_deep_kernel(SG_FWD_PARAMS) { sg_render_fwd_body<true, true>(SG_FWD_ARGS); }
// Frame blockIdx.y of a launch of K frames (sg_common.h, SgBatch): the frame's workspaces at base + frame x stride, its image at
// + frame x 3 H W.  Also the kernel of few-tile frames that share the chip with other views (SG_FLAG_THROUGHPUT), K = 1 included:
// every SIMD has other waves to run there, the pipelined walk's extra vector instructions only cost issue slots (344 -> 319 us for
// the 8 frames of an avatar step, same box).  The single-frame kernels above are the round-3 kernels, untouched: the offsets'
// scalar registers and the weight branch cost the cfg3 forward 1.9 of its 71 us (same-box A/B against the round-3 tree).
// (gridDim.x is a multiple of 8, so blockIdx.x % 8 is the XCD of a workgroup in every frame: sg_tile_of_block.)
template <bool WEIGHTS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
sg_render_fwd_frames_kernel(SgBatch bt, SG_FWD_PARAMS)
{
    const size_t fb = (size_t)blockIdx.y * bt.bin, fg = (size_t)blockIdx.y * bt.geom, fi = (size_t)blockIdx.y * bt.img;
    ranges = sg_at(ranges, fb); pair_keys = sg_at(pair_keys, fb); point_list = sg_at(point_list, fb);
    point_keys = sg_at(point_keys, fb); ck_start = sg_at(ck_start, fb); header = sg_at(header, fb);
    pair_mask = sg_at(pair_mask, fb); tile_count = sg_at(tile_count, fb); long_items = sg_at(long_items, fb);
    plan = sg_at(plan, fb); item_w = sg_at(item_w, fb);
    recA = sg_at(recA, fg); recB = sg_at(recB, fg); recC = sg_at(recC, fg);
    final_T = sg_at(final_T, fi); n_contrib = sg_at(n_contrib, fi); ckpt = sg_at(ckpt, fi);
    out_color += (size_t)blockIdx.y * bt.image;
    sg_render_fwd_body<false, WEIGHTS>(SG_FWD_ARGS);
}

static inline int sg_render_blocks(int T) { return ((T + 8 * SG_XCD_RUN - 1) / (8 * SG_XCD_RUN)) * (8 * SG_XCD_RUN); }

void sg_launch_render_fwd(const SgCam &c, const SgBatch &bt, SgGeom g, SgBin b, size_t cap, SgImg im, float *out_color,
                          int write_keys, hipStream_t st)
{
    static_assert(SG_FB == SG_SEG, "forward batches are the checkpoint granularity");
    const int T = c.gx * c.gy;
    const int grid = sg_render_blocks(T);
    const unsigned K = (unsigned)bt.K;
    const bool few = sg_lds_hist(c.gx, c.gy);
    uint64_t *pk = write_keys ? b.point_keys : (uint64_t *)nullptr;
    const uint32_t key_pitch = sg_key_pitch(c.gx, c.gy, c.flags);   // direct binning: the tile's unsorted keys are its row of tile_keys
    const bool direct = key_pitch != 0u;
#define SG_RF_ARGS(NB) c.W, c.H, c.gx, T, NB, b.ranges, direct ? b.tile_keys : b.pair_keys, b.point_list, pk, g.recA, g.recB, g.recC, c.bg, out_color,     \
                       im.final_T, im.n_contrib, b.ck_start, im.ckpt, sg_ckpt_cap(cap), b.header, b.pair_mask, b.tile_count,       \
                       (const uint4 *)b.sort_items, sg_mask_plane(cap), b.plan, b.item_w, sg_items_cap((size_t)T, cap), key_pitch
    sg_prof_begin(SG_K_RENDER_FWD, st);
    if (K > 1 || (few && (c.flags & SG_FLAG_THROUGHPUT))) {
        // K frames per launch, or a few-tile frame that shares the chip with other views: the plain loop (see the kernel)
        if (few) hipLaunchKernelGGL(sg_render_fwd_frames_kernel<true>, dim3(grid, K), dim3(256), 0, st, bt, SG_RF_ARGS(grid));
        else hipLaunchKernelGGL(sg_render_fwd_frames_kernel<false>, dim3(grid, K), dim3(256), 0, st, bt, SG_RF_ARGS(grid));
    } else if (few) {
        // ONE few-tile frame alone on the chip: the deep kernel.  Upper bound of its extra blocks: quadrants 1..3 of every long
        // tile (most exit at once)
        const int extra = sg_split_long(c.gx, c.gy, c.flags) ? 3 * (int)sg_sort_items_cap(T, cap) : 0;
        hipLaunchKernelGGL(sg_render_fwd_deep_kernel, dim3(extra + grid), dim3(256), 0, st, SG_RF_ARGS(extra));
    } else
        hipLaunchKernelGGL(sg_render_fwd_kernel, dim3(grid), dim3(256), 0, st, SG_RF_ARGS(grid));
    sg_prof_end(SG_K_RENDER_FWD, st);
#undef SG_RF_ARGS
}

// ------------------------------------------------------------------------------------------
#define SG_DPP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, false))

// Sums v[0..8] over the 64 lanes (fixed order).  Afterwards every lane of the 8-lane group g = lane >> 3 holds the
// total of value sg_red_idx(lane); lane 63 additionally gets the total of v[8] in *v8tot.
// v_permlane{32,16}_swap are issued from inline asm: hipcc (ROCm 7.2) folds "r[0] + r[1]" of the builtins' result
// into r[0] + r[0]; the swaps of one level are independent, so one pair of wait states serves all of them.
__device__ __forceinline__ float sg_reduce9(const float v[9], int lane, float *v8tot)
{
    float a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3], a4 = v[4], a5 = v[5], a6 = v[6], a7 = v[7];
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\t"
                 "v_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\ts_nop 1"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    float s01 = a0 + a1, s23 = a2 + a3, s45 = a4 + a5, s67 = a6 + a7;   // (x_lo + x_hi | y_lo + y_hi)
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1"
                 : "+v"(s01), "+v"(s23), "+v"(s45), "+v"(s67));
    float t0 = s01 + s23;                    // rows: v0, v2, v1, v3
    float t1 = s45 + s67;                    // rows: v4, v6, v5, v7
    float u = t0 + SG_DPP(t0, 0x128);        // row_ror:8 -> lanes i, i^8 summed
    float w = t1 + SG_DPP(t1, 0x128);
    float z = (lane & 8) ? w : u;            // per row: lanes 0-7 <- first value, 8-15 <- second
    z += SG_DPP(z, 0xB1);                    // quad_perm [1,0,3,2]
    z += SG_DPP(z, 0x4E);                    // quad_perm [2,3,0,1]
    // row_half_mirror (as asm: the compiler otherwise splits it into a zeroed register, a DPP move and an add sunk into the
    // masked store block)
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(z));
    float x = v[8];
    x += SG_DPP(x, 0xB1); x += SG_DPP(x, 0x4E); x += SG_DPP(x, 0x141); x += SG_DPP(x, 0x140);
    // rows 1 and 3 add the row in front of them, then rows 2 and 3 add lane 31: lane 63 holds the total.  One masked
    // v_add_f32_dpp each (the builtin form costs a zeroed register, a masked move and an add per step)
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa\n\ts_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc" : "+v"(x));
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

// One (entry, quadrant) pass of the backward composite: `ga, gb, gc` the staged record, KPOS the entry's position in the batch
// (compared with the pixel's contributor count inside the batch, `ncq_b`), KSLOT its row in sG.  `continue` skips entries that
// touch no pixel of the quadrant.  Shared by sg_render_bwd_kernel and sg_render_bwd_sparse_kernel.
#define SG_BWD_PASS(KPOS, KSLOT)                                                                                      \
    /* straight-line, predicated (alpha_eff = 0 makes every update an exact no-op) */                                 \
    const float dx = ga.x - pxf, dy = ga.y - pyf;                                                                     \
    const float power = sg_power2(ga.z, ga.w, gb.x, dx, dy);   /* the forward's expression: same decisions */         \
    const float G = __builtin_amdgcn_exp2f(power);                                                                    \
    const float alpha = fminf(0.99f, gb.y * G);                                                                       \
    const bool valid = ((KPOS) < ncq_b) & !(power > 0.0f) & !(alpha < 1.0f / 255.0f);                                 \
    if (__ballot(valid) == 0ull) continue;          /* touches no pixel of this quadrant: the slot stays unset */     \
    const float ae = valid ? alpha : 0.0f;                                                                            \
    const float rinv = __builtin_amdgcn_rcpf(1.0f - ae);      /* rcp(1) == 1 exactly */                               \
    Tr = Tr * rinv;                                  /* T in front of this entry */                                   \
    const float dchan = ae * Tr;                                                                                      \
    /* <colour - colour behind, dL/dpixel>, and the colour behind moves in front of this entry */                     \
    const float e = fmaf(gc, d2, fmaf(gb.w, d1, gb.z * d0)) - Sd;                                                     \
    Sd = fmaf(ae, e, Sd);                                                                                             \
    const float dLa = fmaf(-tb, rinv, e * Tr);      /* + (-T_final / (1 - alpha)) <bg, dL/dpixel> */                  \
    const float w = valid ? G * dLa : 0.0f;          /* = dL/dopacity contribution; dL/dG = o * dLa */                \
    /* first and second moments of w over the pixels: dL/dmean is a per-entry combination of the first moments */     \
    /* (conic . (sum w dx, sum w dy), applied once per record), and so are the factors (-o W/2, -o H/2, -o/2) */      \
    const float wx = w * dx, wy = w * dy;                                                                             \
    float v[9];                                                                                                       \
    v[0] = wx; v[1] = wy;                                                                                             \
    v[2] = wx * dx; v[3] = wx * dy; v[4] = wy * dy;                                                                   \
    v[5] = w;                                                                                                         \
    v[6] = dchan * d0; v[7] = dchan * d1; v[8] = dchan * d2;                                                          \
    float v8;                                                                                                         \
    const float z = sg_reduce9(v, lane, &v8);                                                                         \
    float *slot = &sG[wave][KSLOT][cslot];               /* lanes 0, 8, .., 56: their value's slot; lane 63: slot 8 */\
    if ((lane & 7) == 0) *slot = z;                                                                                   \
    if (lane == 63) *slot = v8;                                                                                       \
    do { } while (0)

// frame blockIdx.y of a launch of K frames (see SG_FWD_FRAME_OFFSETS)
#define SG_BWD_FRAME_OFFSETS                                                                                            \
    {                                                                                                                   \
        const size_t fb = (size_t)blockIdx.y * bt.bin, fg = (size_t)blockIdx.y * bt.geom, fi = (size_t)blockIdx.y * bt.img, \
                     fr = (size_t)blockIdx.y * bt.rec;                                                                   \
        ranges = sg_at(ranges, fb); point_list = sg_at(point_list, fb); header = sg_at(header, fb); items = sg_at(items, fb); \
        ck_start = sg_at(ck_start, fb); pair_mask = sg_at(pair_mask, fb);                                                \
        recA = sg_at(recA, fg); recB = sg_at(recB, fg); recC = sg_at(recC, fg);                                          \
        final_T = sg_at(final_T, fi); n_contrib = sg_at(n_contrib, fi); ckpt = sg_at(ckpt, fi);                          \
        grec_a = sg_at(grec_a, fr); grec_b = sg_at(grec_b, fr);                                                          \
        dL_dpix += (size_t)blockIdx.y * bt.image;                                                                        \
    }
// Round 5: the dependent gathers of an item (list id -> record slot -> records) are done in ONE round per chunk of SG_BCH entries, one
// entry per thread, and the 64-entry batches then run out of LDS -- at cfg3 (lists of ~96 entries) one round per tile instead of one
// per batch; the quadrant-sum slots are re-armed by the combine step itself (one barrier per batch less).  148.4 -> 145.8 us per
// cfg3 view, +1.5 % views/s on one box (A / B / A / B).  Requesting the first chunk's ids and slot words in front of the prologue's
// barrier on top of that: measured, no gain (146.7 us).
#define SG_BCH 128
#ifdef SG_RENDER_STAMP
// diagnostic build only (tools/render_stamps.py): per-workgroup start / end stamps of the shader clock and of the 100 MHz real-time clock
__device__ unsigned long long sg_render_stamps[65536 * 4];
extern "C" int sg_debug_render_stamps(unsigned long long *host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sg_render_stamps), (size_t)n * 8); }
struct SgStampScope {
    unsigned long long c0, r0; unsigned slot;
    __device__ SgStampScope() : c0(__builtin_amdgcn_s_memtime()), r0(__builtin_amdgcn_s_memrealtime()), slot(blockIdx.y * gridDim.x + blockIdx.x) { }
    __device__ ~SgStampScope() {
        if (threadIdx.x == 0 && slot < 65536u) { unsigned long long *q = sg_render_stamps + (size_t)slot * 4; q[0] = c0; q[1] = r0; q[2] = __builtin_amdgcn_s_memtime(); q[3] = __builtin_amdgcn_s_memrealtime(); }
    }
};
#define SG_STAMP_SCOPE SgStampScope sg_stamp_scope_;
#else
#define SG_STAMP_SCOPE
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
sg_render_bwd_kernel(SgBatch bt, int W, int H, int gx, int T, int nblocks, const uint2 *__restrict__ ranges,
                     const uint32_t *__restrict__ point_list, const float4 *__restrict__ recA,
                     const float4 *__restrict__ recB, const float4 *__restrict__ recC,
                     const float *__restrict__ bg, const float *__restrict__ final_T,
                     const uint32_t *__restrict__ n_contrib, const float *__restrict__ dL_dpix,
                     float4 *__restrict__ grec_a, float *__restrict__ grec_b, uint32_t cap, const uint32_t *__restrict__ header,
                     const uint32_t *__restrict__ items, const uint32_t *__restrict__ ck_start,
                     const float4 *__restrict__ ckpt, uint32_t ck_cap, const uint8_t *__restrict__ pair_mask, uint32_t mask_plane,
                     int split_long)
{
    __shared__ float4 sR[SG_BCH][3];           // staged entry: (mean x, mean y, A', B') (C', opacity, colour 0, 1) (colour 2, -, -, -)
    __shared__ uint32_t sM[SG_BCH];
    __shared__ uint16_t sList[4][SG_BB];
    __shared__ float sG[4][SG_BB][9];          // per-quadrant reduced partials of the batch; [8] = SG_UNSET: quadrant w wrote nothing
    __shared__ uint32_t smax[4];
    SG_STAMP_SCOPE
    SG_BWD_FRAME_OFFSETS;
    // one workgroup per work item (tile, depth segment); the item list is in tile order, so the XCD-aware map
    // over the ACTUAL item count keeps neighbouring tiles on one L2.  The grid is an upper bound.
    (void)T; (void)nblocks;
    const int nitems = header[1] ? 0 : (int)header[5];             // nothing to do after a capacity overflow
    const int it = sg_tile_of_block(blockIdx.x);
    if (it >= nitems) return;
    const uint32_t item = items[it];
    const int tile = (int)(item & 0xfffffu), seg = (int)(item >> 20);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tx = tile % gx, ty = tile / gx;
    const int X0 = tx * 16, Y0 = ty * 16;
    const int px = X0 + 8 * (wave & 1) + (lane & 7), py = Y0 + 8 * (wave >> 1) + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    const uint32_t cks = ck_start[tile];
    const int lo = seg * SG_SEG;                                   // this item: entries [lo, hi)
    const int hi = cks != 0xffffffffu && lo + SG_SEG < n ? lo + SG_SEG : n;
    if (lo >= n) return;
    const size_t pid = (size_t)py * W + px, hw = (size_t)H * W;
    // per-pixel state: running T, colour accumulated BEHIND the current entry, dL/dpixel, T_final <bg, dL/dpixel>
    const float Tfin = inside ? final_T[pid] : 0.0f;
    const uint32_t ncq = inside ? n_contrib[pid] : 0u;
    const float d0 = inside ? dL_dpix[pid] : 0.0f, d1 = inside ? dL_dpix[hw + pid] : 0.0f, d2 = inside ? dL_dpix[2 * hw + pid] : 0.0f;
    const float tb = Tfin * (bg[0] * d0 + bg[1] * d1 + bg[2] * d2);
    // (the colour behind enters the gradient only through <S, dL/dpixel>: ONE running scalar Sd instead of three channels)
    float Tr = Tfin, Sd = 0.0f;
    // Not the last segment, and this pixel has contributors behind it: start from the forward checkpoint at `hi`.
    // T in front of entry hi; S = colour composited behind it, normalised by that T.  (A pixel with ncq <= hi has
    // nothing behind: final state.  ncq > hi implies the forward reached that boundary with this pixel live.)
    if (hi < n && ncq > (uint32_t)hi && cks + (uint32_t)(hi / SG_SEG) < ck_cap) {
        const float4 cb = ckpt[(size_t)(cks + (uint32_t)(hi / SG_SEG)) * 256 + tid];
        const float4 cf = ckpt[(size_t)cks * 256 + tid];
        const float rT = 1.0f / cb.x;
        Tr = cb.x;
        Sd = fmaf((cf.w - cb.w) * rT, d2, fmaf((cf.z - cb.z) * rT, d1, (cf.y - cb.y) * rT * d0));
    }
    uint32_t m = ncq;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { uint32_t u = __shfl_xor(m, o, 64); m = u > m ? u : m; }
    const int maxq = (int)__builtin_amdgcn_readfirstlane(m);          // this quadrant's deepest contributor
    if (lane == 0) smax[wave] = m;
    if (tid < SG_BB) {
#pragma unroll
        for (int w = 0; w < 4; w++) sG[w][tid][8] = __uint_as_float(SG_UNSET);
    }
    __syncthreads();
    const int max_contrib = (int)max(max(smax[0], smax[1]), max(smax[2], smax[3]));
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int cslot = lane == 63 ? 8 : sg_red_idx(lane);
    for (int ch = (hi - 1) / SG_BCH; ch >= lo / SG_BCH; ch--) {
        const int cbase = ch * SG_BCH;
        const int ccnt = hi - cbase < SG_BCH ? hi - cbase : SG_BCH;
        // ---- stage the chunk (threads 0..SG_BCH-1): records, quadrant mask, gradient-record slot
        uint32_t rslot = 0xffffffffu;
        float opac = 0.0f, cA = 0.0f, cB = 0.0f, cC = 0.0f;
        if (tid < ccnt) {
            const int e = cbase + tid;
            const uint32_t gid = point_list[range.x + e];
            // quadrants the forward composited this entry in: nothing else can carry a gradient, so neither the
            // rectangle tests nor the two record gathers are repeated for the rest
            // (a long tile of a few-tile frame was composited by four workgroups, one mask plane each: sg_render_fwd_body)
            uint32_t mk = 0u;
            if (e < max_contrib) {
                mk = pair_mask[range.x + e];
                if (split_long && n > SG_WSORT_MAX)
                    mk = (mk & 1u) | (pair_mask[(size_t)mask_plane + range.x + e] & 2u) | (pair_mask[2 * (size_t)mask_plane + range.x + e] & 4u) |
                         (pair_mask[3 * (size_t)mask_plane + range.x + e] & 8u);
            }
            const float4 c4 = recC[SG_REC_STRIDE * (size_t)gid];
            if (mk) {
                const float4 a = recA[SG_REC_STRIDE * (size_t)gid], b = recB[SG_REC_STRIDE * (size_t)gid];
                opac = b.y; cA = a.z; cB = a.w; cC = b.x;
                sR[tid][0] = make_float4(a.x, a.y, SG_KA * a.z, SG_KB * a.w);
                sR[tid][1] = make_float4(SG_KA * b.x, b.y, b.z, b.w);
                sR[tid][2].x = c4.x;
            }
            const uint32_t goff = __float_as_uint(c4.y), mn = __float_as_uint(c4.z), wh = __float_as_uint(c4.w);
            const int x0 = mn & 0xffff, y0 = mn >> 16, rw = wh & 0xffff;
            rslot = goff + (uint32_t)((ty - y0) * rw + (tx - x0));
            sM[tid] = mk;
        }
        __syncthreads();
        for (int kb = (ccnt - 1) / SG_BB; kb >= 0; kb--) {
            const int b0 = kb * SG_BB, base = cbase + b0;                  // the batch inside the chunk / inside the list
            const int cnt = ccnt - b0 < SG_BB ? ccnt - b0 : SG_BB;
            // ---- each wave: the entries that can reach its quadrant, back to front
            if (base < maxq) {
                uint16_t *list = sList[wave];
                const int lim = maxq - base < cnt ? maxq - base : cnt;         // entries >= maxq touch no pixel here
                const int nl = sg_compact_quadrant<1>(sM + b0, lim, wave, lane, lt, list, SG_BB);
                const uint32_t ncq_b = ncq > (uint32_t)base ? ncq - (uint32_t)base : 0u;   // contributors of this pixel inside the batch
                for (int i = nl - 1; i >= 0; i--) {
                    const uint32_t k = list[i];
                    const float4 ga = sR[b0 + k][0], gb = sR[b0 + k][1];
                    const float gc = sR[b0 + k][2].x;
                    SG_BWD_PASS(k, k);
                }
            }
            __syncthreads();
            // ---- combine the quadrants in a fixed order, store the record, re-arm the slots (the thread that staged the entry)
            if (tid >= b0 && tid < b0 + cnt) {
                const int q0 = tid - b0;
                float s[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
                for (int w = 0; w < 4; w++)
                    if (__float_as_uint(sG[w][q0][8]) != SG_UNSET) {
#pragma unroll
                        for (int q = 0; q < 9; q++) s[q] += sG[w][q0][q];
                        sG[w][q0][8] = __uint_as_float(SG_UNSET);
                    }
                if (rslot < cap) {
                    const float no = -opac, nh = 0.5f * no;              // -opacity, -opacity / 2
                    const float m0 = fmaf(cA, s[0], cB * s[1]), m1 = fmaf(cB, s[0], cC * s[1]);   // conic . first moments
                    grec_a[2 * (size_t)rslot] = make_float4(no * ddelx_dx * m0, no * ddely_dy * m1, nh * s[2], nh * s[3]);
                    grec_a[2 * (size_t)rslot + 1] = make_float4(nh * s[4], s[5], s[6], s[7]);
                    grec_b[rslot] = s[8];
                }
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------
// Backward composite for frames of FEW tiles with long lists (an avatar; the regime of sg_lds_hist).
//
// On such a frame the kernel above is not bound by vector issue -- PMC (profiles/r03_avatar_pmc_SQ.csv): 3.1e7 VALU instructions in
// 118 us = 9.4 cycles per instruction on 1024 SIMDs, against 4.1 at cfg3 -- but by what surrounds the arithmetic:
//  * 59 % of the list entries of an avatar frame lie behind saturated pixels.  Their work items still ran four batches of
//    id -> recC gather -> zero record store, because every (tile, Gaussian) record has to exist for the per-Gaussian sums;
//  * a live item paid the dependent chain id -> recC -> mask -> recA / recB once per 64-entry batch, with 64 of its 256 threads.
// Here every record starts INVALID (rec_valid: one byte per record, cleared by the forward's scatter -- rounds 3 streamed 36 B of
// zeros per record in front of every backward instead: 27 MB, 6-10 us per avatar frame) and then
//  * an item whose segment starts behind the tile's deepest contributor returns at once (two loads per thread);
//  * an item stages its whole segment -- 256 entries, one per thread -- in ONE round: id and mask byte first, the three record
//    gathers only for entries the forward composited somewhere (mask != 0); entries with an empty mask are never touched again;
//  * the four 64-entry sub-batches then run out of LDS: compaction, the passes (SG_BWD_PASS: the same arithmetic, bit for bit),
//    the fixed-order combine and ONE record store per (tile, Gaussian) whose mask is not empty, marked valid.
// Results are identical to the kernel above (same passes in the same order; records that kernel writes as zeros are the
// invalid ones, which the per-Gaussian sums take as zeros without loading them: sg_sum_records_coop).
#define SG_BS 256         // entries staged per item (= SG_SEG)
// One workgroup per frame ORDERS the backward work items, heaviest first.  Per-item clocks (round 3): a wave's pass takes ~850 cycles
// whatever else is resident, the heaviest items have ~150 passes per wave (53 us alone), and in list order some of them were
// dispatched 35-47 us into the kernel -- behind 1500 resident items -- and finished at 115 us while the SIMDs idled (3.1e7
// VALU instructions in 118 us = 9.4 cycles per instruction).  Weight = entries of the segment the forward composited anywhere
// (item_w: zeroed by the scatter, written by the forward; a split tile: the largest of its four quadrants); 33 classes,
// counting sort.  The class of each item is kept in LDS between the two passes and the loads are issued eight at a time:
// the block has ~4 k items to place.
#define SG_ITEM_CLASSES 33
__device__ __forceinline__ uint32_t sg_item_class(uint32_t w) { return 32u - (w > 255u ? 32u : (w + 7u) / 8u); }      // 0 = heaviest
__global__ void __launch_bounds__(256)
sg_order_items_kernel(SgBatch bt, const uint32_t *__restrict__ header, const uint32_t *__restrict__ item_w, uint32_t *__restrict__ perm)
{
    {   // frame blockIdx.y (one workgroup per frame)
        const size_t fb = (size_t)blockIdx.y * bt.bin;
        header = sg_at(header, fb); item_w = sg_at(item_w, fb); perm = sg_at(perm, fb);
    }
    {
        constexpr uint32_t KEEP = 8192u;
        __shared__ uint32_t sCls[SG_ITEM_CLASSES];
        __shared__ uint8_t sOf[KEEP];
        const uint32_t nitems = header[1] ? 0u : header[5];
        if (threadIdx.x < SG_ITEM_CLASSES) sCls[threadIdx.x] = 0u;
        __syncthreads();
        for (uint32_t i0 = threadIdx.x; i0 < nitems; i0 += 8u * 256u) {
            uint32_t w[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const uint32_t i = i0 + 256u * u; w[u] = i < nitems ? item_w[i] : 0u; }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t i = i0 + 256u * u;
                if (i < nitems) { const uint32_t cl = sg_item_class(w[u]); if (i < KEEP) sOf[i] = (uint8_t)cl; atomicAdd(&sCls[cl], 1u); }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t run = 0;
            for (int c = 0; c < SG_ITEM_CLASSES; c++) { const uint32_t v = sCls[c]; sCls[c] = run; run += v; }
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nitems; i += 256u) {
            const uint32_t cl = i < KEEP ? (uint32_t)sOf[i] : sg_item_class(item_w[i]);
            perm[atomicAdd(&sCls[cl], 1u)] = i;
        }
    }
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6)))
sg_render_bwd_sparse_kernel(SgBatch bt, int W, int H, int gx, int T, int nblocks, const uint2 *__restrict__ ranges,
                            const uint32_t *__restrict__ point_list, const float4 *__restrict__ recA,
                            const float4 *__restrict__ recB, const float4 *__restrict__ recC,
                            const float *__restrict__ bg, const float *__restrict__ final_T,
                            const uint32_t *__restrict__ n_contrib, const float *__restrict__ dL_dpix,
                            float4 *__restrict__ grec_a, float *__restrict__ grec_b, uint32_t cap, const uint32_t *__restrict__ header,
                            const uint32_t *__restrict__ items, const uint32_t *__restrict__ ck_start,
                            const float4 *__restrict__ ckpt, uint32_t ck_cap, const uint8_t *__restrict__ pair_mask, uint32_t mask_plane,
                            int split_long, const uint32_t *__restrict__ perm, uint8_t *__restrict__ rec_valid)
{
    __shared__ float4 sR[SG_BS][3];            // staged entry: (mean x, mean y, A', B') (C', opacity, colour 0, 1) (colour 2, -, -, -)
    __shared__ uint32_t sM[SG_BS];
    __shared__ uint16_t sList[4][SG_BB];
    __shared__ float sG[4][SG_BB][9];          // per-quadrant reduced partials of the sub-batch; [8] = SG_UNSET: quadrant w wrote nothing
    __shared__ uint32_t smax[4];
    (void)T; (void)nblocks;
    SG_BWD_FRAME_OFFSETS;
    perm = sg_at(perm, (size_t)blockIdx.y * bt.bin); rec_valid = sg_at(rec_valid, (size_t)blockIdx.y * bt.bin);
    const int nitems = header[1] ? 0 : (int)header[5];
    if ((int)blockIdx.x >= nitems) return;
    const uint32_t pit = perm[blockIdx.x];
    const uint32_t item = items[pit];                               // dispatch order = heaviest first
    const int tile = (int)(item & 0xfffffu), seg = (int)(item >> 20);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tx = tile % gx, ty = tile / gx;
    const int X0 = tx * 16, Y0 = ty * 16;
    const int px = X0 + 8 * (wave & 1) + (lane & 7), py = Y0 + 8 * (wave >> 1) + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    const uint32_t cks = ck_start[tile];
    const int lo = seg * SG_SEG;                                   // this item: entries [lo, hi)
    const int hi = cks != 0xffffffffu && lo + SG_SEG < n ? lo + SG_SEG : n;
    if (lo >= n) return;
    const size_t pid = (size_t)py * W + px, hw = (size_t)H * W;
    const uint32_t ncq = inside ? n_contrib[pid] : 0u;
    uint32_t m = ncq;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { uint32_t u = __shfl_xor(m, o, 64); m = u > m ? u : m; }
    const int maxq = (int)__builtin_amdgcn_readfirstlane(m);          // this quadrant's deepest contributor
    if (lane == 0) smax[wave] = m;
    __syncthreads();
    const int max_contrib = (int)max(max(smax[0], smax[1]), max(smax[2], smax[3]));
    if (lo >= max_contrib) return;                                     // nothing of this segment reached a pixel: its records stay zero
    // ---- stage the segment: one entry per thread
    const int cnt = hi - lo;                                           // <= SG_BS
    uint32_t rslot = 0xffffffffu;
    float opac = 0.0f, cA = 0.0f, cB = 0.0f, cC = 0.0f;
    {
        uint32_t mk = 0u, gid = 0u;
        const int e = lo + tid;
        if (tid < cnt && e < max_contrib) {
            gid = point_list[range.x + e];
            mk = pair_mask[range.x + e];
            if (split_long && n > SG_WSORT_MAX)
                mk = (mk & 1u) | (pair_mask[(size_t)mask_plane + range.x + e] & 2u) | (pair_mask[2 * (size_t)mask_plane + range.x + e] & 4u) |
                     (pair_mask[3 * (size_t)mask_plane + range.x + e] & 8u);
            mk &= 15u;
        }
        if (mk) {
            const float4 c4 = recC[SG_REC_STRIDE * (size_t)gid], a = recA[SG_REC_STRIDE * (size_t)gid], b = recB[SG_REC_STRIDE * (size_t)gid];
            const uint32_t goff = __float_as_uint(c4.y), mn = __float_as_uint(c4.z), wh = __float_as_uint(c4.w);
            const int x0 = mn & 0xffff, y0 = mn >> 16, rw = wh & 0xffff;
            rslot = goff + (uint32_t)((ty - y0) * rw + (tx - x0));
            opac = b.y; cA = a.z; cB = a.w; cC = b.x;
            sR[tid][0] = make_float4(a.x, a.y, SG_KA * a.z, SG_KB * a.w);
            sR[tid][1] = make_float4(SG_KA * b.x, b.y, b.z, b.w);
            sR[tid][2].x = c4.x;
        }
        sM[tid] = mk;
    }
    // per-pixel state (requested while the gathers above are in flight)
    const float Tfin = inside ? final_T[pid] : 0.0f;
    const float d0 = inside ? dL_dpix[pid] : 0.0f, d1 = inside ? dL_dpix[hw + pid] : 0.0f, d2 = inside ? dL_dpix[2 * hw + pid] : 0.0f;
    const float tb = Tfin * (bg[0] * d0 + bg[1] * d1 + bg[2] * d2);
    float Tr = Tfin, Sd = 0.0f;
    if (hi < n && ncq > (uint32_t)hi && cks + (uint32_t)(hi / SG_SEG) < ck_cap) {       // (see sg_render_bwd_kernel)
        const float4 cb = ckpt[(size_t)(cks + (uint32_t)(hi / SG_SEG)) * 256 + tid];
        const float4 cf = ckpt[(size_t)cks * 256 + tid];
        const float rT = 1.0f / cb.x;
        Tr = cb.x;
        Sd = fmaf((cf.w - cb.w) * rT, d2, fmaf((cf.z - cb.z) * rT, d1, (cf.y - cb.y) * rT * d0));
    }
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int cslot = lane == 63 ? 8 : sg_red_idx(lane);
    if (tid < SG_BB) {                                                  // armed once; the combine step re-arms what it consumed (round 5:
#pragma unroll                                                          //  one barrier per sub-batch less)
        for (int w = 0; w < 4; w++) sG[w][tid][8] = __uint_as_float(SG_UNSET);
    }
    __syncthreads();
    // ---- the sub-batches, back to front, out of LDS
    for (int sb = (cnt - 1) / SG_BB; sb >= 0; sb--) {
        const int b0 = sb * SG_BB, base = lo + b0;                      // staged index / list position of the sub-batch's first entry
        const int bc = cnt - b0 < SG_BB ? cnt - b0 : SG_BB;
        if (base < maxq) {
            uint16_t *list = sList[wave];
            const int lim = maxq - base < bc ? maxq - base : bc;          // entries >= maxq touch no pixel here
            const int nl = sg_compact_quadrant<1>(sM + b0, lim, wave, lane, lt, list, SG_BB);
            const uint32_t ncq_b = ncq > (uint32_t)base ? ncq - (uint32_t)base : 0u;   // contributors of this pixel inside the sub-batch
            for (int i = nl - 1; i >= 0; i--) {
                const uint32_t k = list[i];
                const float4 ga = sR[b0 + k][0], gb = sR[b0 + k][1];
                const float gc = sR[b0 + k][2].x;
                SG_BWD_PASS(k, k);
            }
        }
        __syncthreads();
        // ---- combine the quadrants in a fixed order and store the record (the thread that staged the entry)
        if (tid >= b0 && tid < b0 + bc) {
            const int q0 = tid - b0;
            float s9[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
            for (int w = 0; w < 4; w++)
                if (__float_as_uint(sG[w][q0][8]) != SG_UNSET) {
#pragma unroll
                    for (int q = 0; q < 9; q++) s9[q] += sG[w][q0][q];
                    sG[w][q0][8] = __uint_as_float(SG_UNSET);
                }
            if (rslot < cap) {
            const float no = -opac, nh = 0.5f * no;
            const float m0 = fmaf(cA, s9[0], cB * s9[1]), m1 = fmaf(cB, s9[0], cC * s9[1]);
            grec_a[2 * (size_t)rslot] = make_float4(no * ddelx_dx * m0, no * ddely_dy * m1, nh * s9[2], nh * s9[3]);
            grec_a[2 * (size_t)rslot + 1] = make_float4(nh * s9[4], s9[5], s9[6], s9[7]);
            grec_b[rslot] = s9[8];
            rec_valid[rslot] = 1;                                  // (every other record of the frame is never read: sg_sum_records_coop)
            }
        }
        __syncthreads();
    }
}

void sg_launch_render_bwd(const SgCam &c, const SgBatch &bt, SgGeom g, SgBin b, size_t cap, SgImg im,
                          const float *dL_dpix, SgRec grec, hipStream_t st)
{
    static_assert(SG_SEG % SG_BB == 0, "segments are whole backward batches");
    const int T = c.gx * c.gy;
    const int grid = sg_render_blocks((int)sg_items_cap((size_t)T, cap));
    const unsigned K = (unsigned)bt.K;
    uint32_t cap32 = cap > 0xffffffffull ? 0xffffffffu : (uint32_t)cap;
    sg_prof_begin(SG_K_RENDER_BWD, st);
    if (sg_lds_hist(c.gx, c.gy)) {
        // few tiles, long lists: order the work items (heaviest first), then only the entries the forward composited are touched --
        // and only their records are marked valid (rec_valid: cleared by the forward's scatter) and ever read
        hipLaunchKernelGGL(sg_order_items_kernel, dim3(1, K), dim3(256), 0, st, bt, b.header, b.item_w, b.item_perm);
        hipLaunchKernelGGL(sg_render_bwd_sparse_kernel, dim3(grid, K), dim3(256), 0, st, bt, c.W, c.H, c.gx, T, grid, b.ranges,
                           b.point_list, g.recA, g.recB, g.recC, c.bg, im.final_T, im.n_contrib, dL_dpix,
                           grec.a, grec.b, cap32, b.header, b.items, b.ck_start, im.ckpt, sg_ckpt_cap(cap), b.pair_mask, sg_mask_plane(cap),
                           sg_split_long(c.gx, c.gy, c.flags) ? 1 : 0, b.item_perm, b.rec_valid);
    } else
    hipLaunchKernelGGL(sg_render_bwd_kernel, dim3(grid, K), dim3(256), 0, st, bt, c.W, c.H, c.gx, T, grid, b.ranges,
                       b.point_list, g.recA, g.recB, g.recC, c.bg, im.final_T, im.n_contrib, dL_dpix,
                       grec.a, grec.b, cap32, b.header, b.items, b.ck_start, im.ckpt, sg_ckpt_cap(cap), b.pair_mask, sg_mask_plane(cap),
                       sg_split_long(c.gx, c.gy, c.flags) ? 1 : 0);
    sg_prof_end(SG_K_RENDER_BWD, st);
}
