// Shared device/host helpers for libsings_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/sings_hip.h"

#define SG_WAVE 64
// (The compile-time experiment switches of rounds 1-2 -- SG_EXP bits that compiled phases of a kernel out for timing -- are no
// longer in the product sources; the measurements they produced are in LAB.md, the variants in the git history.)
#ifdef SG_EXP
#error "SG_EXP experiment variants were removed from the product sources"
#endif
// backward work items are tile | depth segment << 20: images of 2^20 tiles or more (> 16k x 16k pixels) are rejected by
// sg_layout / every entry point instead of aliasing tile ids
#define SG_MAX_TILES (1u << 20)
// Tile lists longer than SG_SEG entries are cut into depth segments of SG_SEG entries: the forward checkpoints the
// per-pixel (T, colour) at every segment boundary, the backward runs one workgroup per (tile, segment).
#define SG_SEG 256
// number of segments (= checkpoint slots) of a tile list of n entries; 0: not segmented (segment id has 12 bits)
__host__ __device__ inline uint32_t sg_nseg(uint32_t n) { return n > SG_SEG && n <= SG_SEG * 4095u ? (n + SG_SEG - 1) / SG_SEG : 0u; }
// capacities of the work-item list and of the checkpoint buffer (slots of 256 float4) for T tiles / cap pairs
__host__ __device__ inline uint32_t sg_items_cap(size_t T, size_t cap) { size_t v = T + cap / SG_SEG + 2; return v > 0xffffffffull ? 0xffffffffu : (uint32_t)v; }
// Long lists (> 1024 entries: what a compositing workgroup does not sort itself) are bucket-sorted (sg_binning.hip): one
// partition work item per such tile, and per tile a block of reserved group slots.  A list of n keys needs at most
// 5 n / 1024 + 1 groups (greedy packing: two neighbouring groups hold > 1024 keys; buckets that are split again break runs).
__host__ __device__ inline uint32_t sg_group_slots(uint32_t n) { return n > 1024u ? 6u * ((n + 1023u) / 1024u) + 2u : 0u; }
__host__ __device__ inline uint32_t sg_sort_items_cap(size_t T, size_t cap) { size_t v = (T < cap / 1024 ? T : cap / 1024) + 2; return v > 0xffffffffull ? 0xffffffffu : (uint32_t)v; }
// sum of sg_group_slots over the long tiles: 6 (n / 1024 + 1) + 2 <= 14 n / 1024 for n > 1024
__host__ __device__ inline uint32_t sg_rank_items_cap(size_t cap) { size_t v = cap / 73 + 16; return v > 0xffffffffull ? 0xffffffffu : (uint32_t)v; }
__host__ __device__ inline uint32_t sg_ckpt_cap(size_t cap) { size_t v = 2 * (cap / SG_SEG) + 2; return v > 0xffffffffull ? 0xffffffffu : (uint32_t)v; }
// Tile counters: one packed word per tile (a Gaussian's tiles are neighbours, so one 64-lane atomic instruction
// touches few lines).  When the image has few tiles (<= SG_HIST_TILES_MAX) a single tile can collect ~1e4 pairs and a
// counter word sustains only ~88 returning atomics per microsecond; there the preprocess kernels count a workgroup's
// pairs per tile in an LDS histogram first and issue ONE global atomic per (workgroup, touched tile).  (Earlier
// schemes for that regime -- one counter per 128-B line, then 8 sub-counters per tile -- took 141 / 80 us of
// preprocess + 22 us of scan on the avatar frame; the histogram: 43 + 7 us.)
// Tile counters live in 4x4 BLOCKS of tiles: the 16 counters of a block share one 64-byte line.  A Gaussian's rectangle is a few
// neighbouring tiles in x AND y, so the returning atomics of its pairs -- one per (tile, Gaussian), 780 k per cfg3 view, the
// largest part of the preprocess -- touch one or two lines instead of one per tile row (row-major: 18.0 us for the access pattern
// of a cfg3 view, this layout 14.9 us; round-2 replay, adopted in round 3).  Counters of tiles outside the image are never touched.
__host__ __device__ inline uint32_t sg_ctr_blocks_x(uint32_t gx) { return (gx + 3u) >> 2; }
__host__ __device__ inline size_t sg_ctr_count(uint32_t gx, uint32_t gy) { return (size_t)sg_ctr_blocks_x(gx) * ((gy + 3u) >> 2) * 16u; }
__host__ __device__ inline uint32_t sg_ctr_index(uint32_t tx, uint32_t ty, uint32_t gx)
{
    return ((ty >> 2) * sg_ctr_blocks_x(gx) + (tx >> 2)) * 16u + ((ty & 3u) << 2) + (tx & 3u);
}
__host__ __device__ inline uint32_t sg_ctr_of_tile(uint32_t tile, uint32_t gx) { return sg_ctr_index(tile % gx, tile / gx, gx); }

#define SG_HIST_TILES_MAX 4096
// (the histogram is indexed like the counters: sg_ctr_index)
static inline bool sg_lds_hist(int gx, int gy) { return sg_ctr_count((uint32_t)gx, (uint32_t)gy) <= SG_HIST_TILES_MAX; }
// The same regime (few tiles, a handful of them with lists of thousands of entries) is where the forward composite ends in a few
// deep tiles running alone: there a tile of more than 1024 entries is composited by four workgroups, one per quadrant
// (sg_render.hip), and the per-entry quadrant masks live in four planes of mask_plane bytes.
// DIRECT binning: the caller vouches for lists of <= SG_WSORT_MAX entries (SG_FLAG_SHORT_LISTS, checked on the device) and the image has
// too many tiles for the per-workgroup histogram -- then a pair's key goes straight into its tile's row of SgBin::tile_keys at the rank
// its counting atomic returned: no pair records, no scatter pass (round 6; sg_project.h::sg_store_proj, sg_binning.hip)
#define SG_TILE_KEY_CAP 1024        // many tiles: keys a row holds = SG_WSORT_MAX (sg_sort.h) = entries between rows
#define SG_TILE_KEY_PITCH 1024
#define SG_TILE_ROW_LONG 16384      // few tiles (SG_FLAG_LONG_ROWS): keys a row holds = entries between rows
// entries between the rows of SgBin::tile_keys for this frame; 0: plain binning (pair records + scatter pass)
__host__ __device__ inline uint32_t sg_key_pitch_of(bool hist, int flags)
{
#ifdef SG_NO_DIRECT      /* A/B builds only (tools/ab_direct.sh) */
    (void)hist; (void)flags;
    return 0u;
#else
    if (hist) return (flags & SG_FLAG_LONG_ROWS) ? (uint32_t)SG_TILE_ROW_LONG : 0u;
    return (flags & SG_FLAG_SHORT_LISTS) ? (uint32_t)SG_TILE_KEY_PITCH : 0u;
#endif
}
static inline uint32_t sg_key_pitch(int gx, int gy, int flags) { return sg_key_pitch_of(sg_lds_hist(gx, gy), flags); }
static inline bool sg_split_long(int gx, int gy, int flags) { return sg_lds_hist(gx, gy) && !(flags & SG_FLAG_THROUGHPUT); }
static inline uint32_t sg_mask_plane(size_t cap) { const size_t v = (cap + 256) & ~(size_t)255; return v > 0xffffffffull ? 0xffffffffu : (uint32_t)v; }

// float4 units between the geometry records of consecutive Gaussians: recA / recB / recC interleaved + 16 B of padding = one 64-B line
#define SG_REC_STRIDE (SG_GEOM_REC_BYTES / 16)
struct SgGeom {            // per-Gaussian projected records: recA / recB / recC point into ONE array of 64-B records (SG_REC_STRIDE)
    float4 *recA;          // (pix.x, pix.y, conic.x, conic.y)
    float4 *recB;          // (conic.z, opacity, r, g)
    float4 *recC;          // (b, bits(goff), bits(minx | miny<<16), bits(w | h<<16))
    float *depth;          // view-space z
    uint32_t *flags;       // bits 0..2: SH colour channel clamped at 0
    uint2 *slot;           // (first gradient-record slot, rect width | height << 16): recC.y / recC.w again, dense, for the per-Gaussian backward
};

struct SgBin {
    uint32_t *header;      // [0] R  [1] overflow  [2] pair allocator  [3] ntiles  [4] sort items  [5] backward work items  [6] rank items
    uint32_t *tile_count;  // [sg_ctr_count(gx, gy)] pairs per tile, indexed by sg_ctr_index
    uint2 *ranges;         // [T] (start,end) into point_list
    uint32_t *cursor;      // [T] unclipped first slot of every tile
    uint64_t *pair_keys;   // [cap] (depth_bits << 32 | gid), grouped by tile, sorted in place
    uint32_t *point_list;  // [cap] sorted Gaussian ids
    uint64_t *point_keys;  // [cap] optional upstream-format keys
    uint32_t *pair_gid;    // [cap] Gaussian-major pair list written by the preprocess: Gaussian id,
    uint32_t *pair_tile;   //       tile id,
    uint32_t *pair_local;  //       arrival rank inside the tile (returned by the counting atomic)
    uint4 *sort_items;     // (tile, first group slot, first entry, entries) of every list of more than 1024 entries; count in header[4]
    uint2 *rank_items;     // group slots of the bucket sort: (first index | buffer flag, length | tile << 11); reserved count in header[6]
    uint32_t *items;       // backward work items: tile | segment << 20; count in header[5]
    uint32_t *ck_start;    // [T] first checkpoint slot of a segmented tile
    uint4 *plan;           // [T] (first backward item, first sort item, first rank item, pair count)
    uint8_t *pair_mask;    // [cap] per sorted list entry: quadrants the forward composited it in (0 if it never staged it)
    uint32_t *item_w;      // [items_cap] weight of a backward work item: the scatter zeroes it, the few-tile forward sets it to the entries
                           //             of the segment it composited somewhere (split tiles: the largest of the four quadrants)
    uint32_t *item_perm;   // [items_cap]    ... and the work items by descending weight (sg_order_items_kernel)
    uint64_t *tile_keys;   // [T][SG_TILE_KEY_PITCH] direct binning (sg_direct_keys): the unsorted keys of tile t in row t, written by the preprocess
    uint8_t *rec_valid;    // [cap] few-tile frames: 1 = the sparse backward composite wrote the gradient record of this Gaussian-major
                           //       pair slot; zeroed by the forward's scatter (one lane per pair anyway).  Readers skip the others:
                           //       the 36-B records themselves are never zeroed (round 3 streamed 27 MB of zeros per avatar frame)
};

struct SgImg {
    float *final_T;        // [H*W]
    uint32_t *n_contrib;   // [H*W]
    float4 *ckpt;          // [slots][256 pixels] (T, C.r, C.g, C.b); slot 0 of a tile = final state
};

static inline size_t sg_align(size_t x) { return (x + 255) & ~(size_t)255; }

// Backward workspace: one 36-byte gradient record per (tile, Gaussian) pair -- (d mean2D.x, d mean2D.y, d conic.x, d conic.y)
// (d conic.w, d opacity, d colour r, g) as two 16-byte vectors in plane `a`, d colour b as one float in plane `b`.  (Rounds 1-2:
// three 16-byte vectors, the last one 3/4 padding: 12 B per pair written and read for nothing, 18.7 MB per cfg3 view.)
struct SgRec { float4 *a; float *b; };
static inline size_t sg_rec_bytes(size_t cap) { return sg_align((cap + 1) * 32) + sg_align((cap + 1) * 4 + 16); }     // (+16: plane b is zeroed in 16-byte stores)
static inline SgRec sg_rec_view(const void *bwd_ws, size_t cap)
{
    SgRec r;
    r.a = (float4 *)bwd_ws; r.b = (float *)((char *)bwd_ws + sg_align((cap + 1) * 32));
    return r;
}

static inline SgGeom sg_geom_view(void *ws, const SgLayout &L)
{
    char *b = (char *)ws;
    SgGeom g;
    g.recA = (float4 *)(b + L.geom_recA); g.recB = (float4 *)(b + L.geom_recB);
    g.recC = (float4 *)(b + L.geom_recC); g.depth = (float *)(b + L.geom_depth);
    g.flags = (uint32_t *)(b + L.geom_flags); g.slot = (uint2 *)(b + L.geom_slot);
    return g;
}
static inline SgBin sg_bin_view(void *ws, const SgLayout &L)
{
    char *b = (char *)ws;
    SgBin g;
    g.header = (uint32_t *)(b + L.bin_header); g.tile_count = (uint32_t *)(b + L.bin_tile_count);
    g.ranges = (uint2 *)(b + L.bin_ranges); g.cursor = (uint32_t *)(b + L.bin_cursor);
    g.pair_keys = (uint64_t *)(b + L.bin_pair_keys); g.point_list = (uint32_t *)(b + L.bin_point_list);
    g.point_keys = (uint64_t *)(b + L.bin_point_keys);
    g.pair_gid = (uint32_t *)(b + L.bin_pair_gid); g.pair_tile = (uint32_t *)(b + L.bin_pair_tile);
    g.pair_local = (uint32_t *)(b + L.bin_pair_local);
    g.sort_items = (uint4 *)(b + L.bin_sort_items); g.rank_items = (uint2 *)(b + L.bin_rank_items);
    g.items = (uint32_t *)(b + L.bin_items); g.ck_start = (uint32_t *)(b + L.bin_ck_start);
    g.plan = (uint4 *)(b + L.bin_plan); g.pair_mask = (uint8_t *)(b + L.bin_pair_mask);
    g.item_w = (uint32_t *)(b + L.bin_item_w); g.item_perm = (uint32_t *)(b + L.bin_item_perm);
    g.rec_valid = (uint8_t *)(b + L.bin_rec_valid); g.tile_keys = (uint64_t *)(b + L.bin_tile_keys);
    return g;
}
static inline SgImg sg_img_view(void *ws, const SgLayout &L)
{
    char *b = (char *)ws;
    SgImg g;
    g.final_T = (float *)(b + L.img_final_T); g.n_contrib = (uint32_t *)(b + L.img_n_contrib);
    g.ckpt = (float4 *)(b + L.img_ckpt);
    return g;
}

// ---- K frames per launch (round 4) -------------------------------------------------------------------------------------------
// The frames (poses / cameras) of one optimisation step render the SAME canonical Gaussians: sings_hybrid.py:474-569 hands the model
// 16 frames per call, and the frame-parallel step renders K frames per rank.  Every kernel of the path takes a frame index --
// blockIdx.y (composite / binning / loss kernels), a decoded linear block id (per-Gaussian forward: the K blocks of one Gaussian
// block run back to back on ONE XCD, so the canonical inputs come from HBM once and from that XCD's L2 K - 1 times), or an in-kernel
// loop (per-Gaussian backward: the K frames' gradients are summed in registers, the gradient row is written ONCE) -- and finds the
// frame's workspaces at base + frame * stride: K consecutive workspaces of sg_layout's sizes in one allocation.  K = 1 with zero
// strides is the single-frame call: the same kernels, bit for bit.
#ifndef SG_MAX_FRAMES
#define SG_MAX_FRAMES 16
#endif
struct SgBatch {
    int K;                         // frames in this launch
    int cam_stride;                // 0: one camera for all frames; 1: frame f reads view + 16 f, proj + 16 f, campos + 3 f
    int transl_stride;             // floats between the frames' translations (0 or 3)
    int P;                         // Gaussians (radii / means2D of frame f at + f P)
    size_t geom, bin, img, rec;    // bytes between consecutive frames' workspaces
    size_t image;                  // floats of one [3,H,W] image (out_color / dL_dout_color of frame f at + f image)
};
static inline SgBatch sg_batch_one(int P) { SgBatch b; b.K = 1; b.cam_stride = 0; b.transl_stride = 0; b.P = P; b.geom = b.bin = b.img = b.rec = 0; b.image = 0; return b; }
// Linear workgroup id -> (Gaussian block, frame) for the per-Gaussian FORWARD kernels.  Workgroups are dealt round-robin to the 8
// XCDs (id % 8); inside an XCD the K frames of one Gaussian block take consecutive slots, so the block's canonical inputs (means,
// scales, SH rows, 4 J bytes of skinning weights per Gaussian) are fetched from HBM by the first of them and served by that XCD's
// L2 to the others.  K = 1: the identity.  Grid: sg_frame_grid(nblocks, K) workgroups; false = surplus workgroup.
__device__ __forceinline__ bool sg_block_frame(int id, int K, int nblocks, int &gblock, int &frame)
{
    const int xcd = id & 7, slot = id >> 3;
    frame = slot % K;
    gblock = (slot / K) * 8 + xcd;
    return gblock < nblocks;
}
static inline unsigned sg_frame_grid(int nblocks, int K) { return (unsigned)(((nblocks + 7) / 8) * 8 * K); }
template <class T> __host__ __device__ __forceinline__ T *sg_at(T *p, size_t bytes) { return p ? (T *)((char *)p + bytes) : p; }
__host__ __device__ __forceinline__ SgGeom sg_frame(SgGeom g, size_t off)
{
    g.recA = sg_at(g.recA, off); g.recB = sg_at(g.recB, off); g.recC = sg_at(g.recC, off);
    g.depth = sg_at(g.depth, off); g.flags = sg_at(g.flags, off); g.slot = sg_at(g.slot, off);
    return g;
}
__host__ __device__ __forceinline__ SgBin sg_frame(SgBin b, size_t off)
{
    b.header = sg_at(b.header, off); b.tile_count = sg_at(b.tile_count, off); b.ranges = sg_at(b.ranges, off);
    b.cursor = sg_at(b.cursor, off); b.pair_keys = sg_at(b.pair_keys, off); b.point_list = sg_at(b.point_list, off);
    b.point_keys = sg_at(b.point_keys, off); b.pair_gid = sg_at(b.pair_gid, off); b.pair_tile = sg_at(b.pair_tile, off);
    b.pair_local = sg_at(b.pair_local, off); b.sort_items = sg_at(b.sort_items, off); b.rank_items = sg_at(b.rank_items, off);
    b.items = sg_at(b.items, off); b.ck_start = sg_at(b.ck_start, off); b.plan = sg_at(b.plan, off);
    b.pair_mask = sg_at(b.pair_mask, off); b.item_w = sg_at(b.item_w, off); b.item_perm = sg_at(b.item_perm, off);
    b.rec_valid = sg_at(b.rec_valid, off); b.tile_keys = sg_at(b.tile_keys, off);
    return b;
}
__host__ __device__ __forceinline__ SgImg sg_frame(SgImg i, size_t off)
{
    i.final_T = sg_at(i.final_T, off); i.n_contrib = sg_at(i.n_contrib, off); i.ckpt = sg_at(i.ckpt, off);
    return i;
}
__host__ __device__ __forceinline__ SgRec sg_frame(SgRec r, size_t off) { r.a = sg_at(r.a, off); r.b = sg_at(r.b, off); return r; }

// Kernel parameter block for per-Gaussian kernels
struct SgCam {
    int W, H, gx, gy, flags;
    float tanfovx, tanfovy, fx, fy, mod;
    int D, M;
    const float *view, *proj, *campos, *bg;
    unsigned long long *count_signal;      // mapped host word the tile scan publishes (valid bit | flags << 32 | R) to, or null
};
__host__ __device__ __forceinline__ SgCam sg_frame(SgCam c, int frame, int cam_stride)
{
    const size_t f = (size_t)frame * (size_t)cam_stride;
    c.view += 16 * f; c.proj += 16 * f; c.campos += 3 * f;
    if (c.count_signal) c.count_signal += frame;
    return c;
}

// launchers (defined in the .hip files)
void sg_launch_preprocess_fwd(const SgCam &c, const SgBatch &bt, int P, const float *means3D, const float *shs,
                              const float *colors_precomp, const float *opacities, const float *scales,
                              const float *rotations, const float *cov3D_precomp, SgGeom g, SgBin b,
                              size_t cap, int32_t *radii, hipStream_t st);
void sg_launch_binning(const SgCam &c, const SgBatch &bt, int P, const int32_t *radii, SgGeom g, SgBin b, size_t cap,
                       int write_keys, hipStream_t st);
static inline uint32_t sg_cap32(size_t cap) { return cap > 0xffffffffull ? 0xffffffffu : (uint32_t)cap; }
void sg_launch_render_fwd(const SgCam &c, const SgBatch &bt, SgGeom g, SgBin b, size_t cap, SgImg im, float *out_color,
                          int write_keys, hipStream_t st);
void sg_launch_render_bwd(const SgCam &c, const SgBatch &bt, SgGeom g, SgBin b, size_t cap, SgImg im,
                          const float *dL_dpix, SgRec grec, hipStream_t st);
// a9out [K][P][12]: the per-Gaussian sums of frame f's gradient records (sg_skin.hip::sg_record_sums_kernel), for the K-frame
// per-Gaussian backward kernels
void sg_launch_record_sums(const SgBatch &bt, int P, const int32_t *radii, SgGeom g, SgRec grec, size_t cap, const uint32_t *header,
                           const uint8_t *rec_valid, float4 *a9out, hipStream_t st);
// bytes behind the K record buffers of a K-frame backward workspace: the a9 block of the K-camera per-Gaussian backward (K > 1)
static inline size_t sg_a9_bytes(int P, int K) { return K > 1 ? sg_align((size_t)K * (size_t)P * 48 + 256) : 0; }
void sg_launch_preprocess_bwd(const SgCam &c, const SgBatch &bt, int P, const float *means3D, const float *shs,
                              const float *colors_precomp, const float *opacities, const float *scales,
                              const float *rotations, const float *cov3D_precomp,
                              const int32_t *radii, SgGeom g, SgRec grec, size_t cap, const uint32_t *header, const uint8_t *rec_valid,
                              float4 *a9buf, float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dsh,
                              float *dL_dcolors, float *dL_dopacity, float *dL_dscales,
                              float *dL_drots, float *dL_dcov3D, int accumulate, hipStream_t st);

// per-kernel event timing (sg_api.hip)
// Zero `bytes` (a multiple of 4) at the 4-byte aligned address p with a KERNEL.  Not hipMemsetAsync: a step replayed from
// a HIP graph went wrong on ROCm 7.2 as soon as the host had synchronised once between two replays when the graph
// contained memset nodes (tools/graph_gap.py reproduces it); a kernel node has no such problem and is no slower.
void sg_zero_async(void *p, size_t bytes, hipStream_t st);
void sg_launch_joint_transforms(int B, int J, const float *pose, const float *joints, const int *parents, const float *post,
                                const float *dA, float *out0, float *out1, hipStream_t st);
void sg_launch_m2q(int N, const float *m, const float *dq, float *out, hipStream_t st);
int sg_launch_rot_map(int op, int N, const float *in, const float *g, float *out, hipStream_t st);
void sg_launch_qmul(int N, const float *a, const float *b, const float *g, float *out0, float *out1, hipStream_t st);
void sg_launch_lbs_fwd(int P, int J, const float *W, const float *A, const float *v, float *T_out, float *verts, hipStream_t st);
void sg_launch_lbs_bwd(int P, int J, const float *W, const float *A, const float *v, const float *dT, const float *dverts,
                       float *slab, float *dv, float *dA, hipStream_t st);
void sg_prof_begin(int id, hipStream_t st);
void sg_prof_end(int id, hipStream_t st);

// LBS-fused per-Gaussian kernels (sg_skin.hip)
void sg_launch_skin_fwd(const SgCam &c, const SgBatch &bt, int P, const SgSkinInputs *in, const float *shs, const float *opacities,
                        const float *scales, SgGeom g, SgBin b, size_t cap, int32_t *radii, float *posed_xyz,
                        float *posed_rotq, float *posed_scales, hipStream_t st);
size_t sg_skin_slab_floats(int P, int K);
size_t sg_photo_loss_ws_bytes_impl(int W, int H);
int sg_tp_check(const SgTriplane *tp);
size_t sg_triplane_ws_bytes_impl(const SgTriplane *tp);
size_t sg_triplane_bwd_ws_bytes_impl(const SgTriplane *tp, int N);
void sg_launch_triplane_fwd(const SgTriplane *tp, int N, const float *xyz, void *ws, float *feats, hipStream_t st);
int sg_launch_triplane_bwd(const SgTriplane *tp, int N, const float *xyz, void *ws, const float *dfeats,
                           float *const dplanes[4][3], float *dxyz, hipStream_t st);
int sg_launch_triplane_bwd_prepare(const SgTriplane *tp, int N, const float *xyz, void *ws, hipStream_t st);
int sg_launch_triplane_bwd_run(const SgTriplane *tp, int N, const float *xyz, void *ws, const float *dfeats,
                           float *const dplanes[4][3], float *dxyz, hipStream_t st);
size_t sg_bias_act_ws_bytes_impl(int N, int C);
void sg_launch_bias_act_fwd(int N, int C, int act, const float *y, const float *bias, const float *row_offset,
                            float *z_out, float *h_out, hipStream_t st);
void sg_launch_bias_act_bwd(int N, int C, int act, const float *z, const float *row_offset, const float *dh, void *ws,
                            float *dz, float *dbias, hipStream_t st);
size_t sg_weight_grad_ws_bytes_impl(int N, int Cout, int Cin);
int sg_launch_linear_fwd(int N, int Cin, int Cout, int act, const float *x, const float *W, const float *bias,
                         const float *row_offset, float *z_out, float *h_out, hipStream_t st);
int sg_launch_linear_bwd(int N, int Cin, int Cout, int act, const float *z, const float *row_offset, const float *dh,
                         const float *W, float *dz_out, float *dx_out, hipStream_t st, int accumulate);
int sg_launch_weight_grad(int N, int Cout, int Cin, const float *dz, const float *x, void *ws, float *dW, float *db,
                          hipStream_t st);
size_t sg_reg_ws_bytes_impl(int n);
size_t sg_knn_ws_bytes_impl(int N);
void sg_launch_region_laplacian(int V, int C, const float *x, const int *row_ptr, const int *col, const float *deg_inv,
                                const float *vscale, void *ws, float *g_ws, float *loss, const float *upstream,
                                float *dL_dx, hipStream_t st);
void sg_launch_mesh_edge(int V, int E, const float *x, const int *row_ptr, const int *col, void *ws, float *loss,
                         const float *upstream, float *dL_dx, hipStream_t st);
void sg_launch_l2norm(int N, const float *off, const float *scales, const float *opacity, const float *lambdas6, void *ws,
                      float *loss, const float *upstream, float *d_off, float *d_scales, float *d_opacity, hipStream_t st);
void sg_launch_knn_prepare(int N, const float *xyz, void *ws, hipStream_t st);
int sg_launch_knn_finish(int N, int K, const float *scales, void *ws, float *mean_edge_out, float *loss, const float *upstream,
                         float *d_scales, hipStream_t st);
int sg_launch_knn_edge(int N, int K, const float *xyz, const float *scales, void *ws, float *mean_edge_out, float *loss,
                       const float *upstream, float *d_scales, hipStream_t st);
void sg_launch_photo_loss(int K, int W, int H, float l1_w, float ssim_w, const float *raw, const float *gt_rgb,
                          const float *mask, const float *bg, void *ws, float *pred_out, float *gt_out,
                          float *losses, const float *upstream, float *dL_draw, size_t gt_stride, size_t mask_stride, hipStream_t st);
void sg_launch_photo_loss_bwd(int K, int W, int H, float l1_w, float ssim_w, const float *raw, const float *gt_rgb,
                              const float *mask, const float *bg, const void *ws, const float *upstream, float *dL_draw,
                              size_t gt_stride, size_t mask_stride, hipStream_t st, int up_stride = 0);
void sg_launch_skin_bwd(const SgCam &c, const SgBatch &bt, int P, const SgSkinInputs *in, const float *shs, const float *scales,
                        const int32_t *radii, SgGeom g, SgRec grec, size_t cap, const uint32_t *header, const uint8_t *rec_valid,
                        const float *dposed_xyz_in, const float *dposed_rotq_in, float *slab, float *dL_dxyz_canon, float *dL_drot_canon,
                        float *dL_dscales, float *dL_dopacity, float *dL_dsh, float *dL_dmeans2D, float *dL_dA,
                        float *dL_dtransl, int accumulate, hipStream_t st);
