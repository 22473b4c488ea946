/*
 * sings_hip.h -- C ABI of libsings_hip.so, the MI355X (gfx950) render path for SinGS.
 *
 * This library replaces the pybind11 module `diff_gaussian_rasterization._C` that the
 * reference reaches through
 *     sings/rec/renderer/gs_renderer_single.py:6-9,69-95   (GaussianRasterizer call)
 *     sings/rec/renderer/gs_renderer_multiple.py:6-9,95-121
 * (the package itself is the un-vendored dependency of install_all.sh:22), and adds fused
 * entry points for the SMPL-LBS deformation block of
 *     sings/rec/models/sings_hybrid.py:398-428 / sings/rec/utils/body_model/lbs.py:59-74.
 *
 * Contract
 *  - plain C: raw DEVICE pointers (fp32 / int32, contiguous), sizes, a hipStream_t passed
 *    as void*.  No torch types.  The caller owns every byte (inputs, outputs, workspaces).
 *  - every entry point returns 0 on success, non-zero on error; sg_last_error() gives the
 *    message (thread-local).
 *  - kernels are enqueued on the stream given; nothing synchronises unless stated.
 *  - re-entrant; no global mutable state besides the thread-local error string.
 */
#ifndef SINGS_HIP_H
#define SINGS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SG_TILE 16            /* BLOCK_X = BLOCK_Y of the upstream rasterizer */
#define SG_GRAD_REC_FLOATS 9  /* floats per (tile,Gaussian) gradient record: two 16-B vectors + one float in a separate plane */

/* Mirrors the 12-field GaussianRasterizationSettings NamedTuple the reference builds at
 * gs_renderer_single.py:69-82 (bg/viewmatrix/projmatrix/campos are device pointers). */
typedef struct SgRasterSettings {
    int32_t image_height;
    int32_t image_width;
    float tanfovx;
    float tanfovy;
    float scale_modifier;
    int32_t sh_degree;     /* active degree D (0..3) */
    int32_t sh_coeffs;     /* M: rows allocated per Gaussian in `shs` ([P,M,3]) */
    int32_t prefiltered;
    int32_t debug;         /* 1: synchronise + check after every kernel */
    int32_t flags;         /* SG_FLAG_* (0 = none) */
    const float *bg;         /* [3]  */
    const float *viewmatrix; /* [16] row-major torch tensor = column-major matrix */
    const float *projmatrix; /* [16] */
    const float *campos;     /* [3]  */
    /* Optional early pair count (both NULL = off).  `count_signal` is the DEVICE address, `count_signal_host` the host address of
     * ONE 64-bit word of pinned, mapped, coherent host memory (sg_signal_alloc).  A forward called with num_rendered_host != NULL
     * clears the word, and its binning kernel stores (1 << 63 | flags << 32 | R) there the moment the pair count exists -- after
     * the per-Gaussian kernel and the tile scan, a third of the way into a cfg3 forward -- so the call returns R WITHOUT
     * synchronising the stream: the composite kernel is still running while the caller already queues its next work.  This is
     * what lets the drop-in autograd wrapper keep upstream's guarantee (upstream blocks the host mid-forward for R: never a
     * frame rendered with a too small workspace) at nearly the speed of never looking.
     * flags: bit 0 more pairs than the workspace holds, bit 1 a long list under SG_FLAG_SHORT_LISTS (both: nothing was composited),
     * bit 2 (SG_COUNT_FLAG_HALF_ROWS, in this word only) no tile list is longer than 512 entries -- a caller that reads the word
     * itself after the call may pass SG_FLAG_SHORT_LISTS for the next frame of the scene (the Python wrapper does); bit 3
     * (SG_COUNT_FLAG_HALF_LONG_ROWS) none longer than 8192 -- the same for SG_FLAG_LONG_ROWS. */
    unsigned long long *count_signal;
    volatile unsigned long long *count_signal_host;
} SgRasterSettings;

/* Byte offsets inside the opaque workspaces (exposed for tests / debugging only). */
#define SG_GEOM_REC_BYTES 64
typedef struct SgLayout {
    /* geometry workspace (per Gaussian).  recA / recB / recC: three 16-byte vectors INTERLEAVED in one 64-byte record per
     * Gaussian (SG_GEOM_REC_BYTES): vector k of Gaussian i is at geom_rec{A,B,C} + 64 i */
    size_t geom_recA, geom_recB, geom_recC, geom_depth, geom_flags, geom_slot, geom_bytes;
    /* binning workspace */
    size_t bin_header, bin_tile_count, bin_ranges, bin_cursor, bin_pair_keys, bin_point_list,
        bin_point_keys, bin_pair_gid, bin_pair_tile, bin_pair_local,
        bin_sort_items, bin_rank_items, bin_items,      /* work lists: long-list sort chunks, chunk merges, backward segments */
        bin_ck_start, bin_plan, bin_pair_mask, bin_item_w, bin_item_perm,
        bin_rec_valid,          /* [cap] one byte per gradient record (Gaussian-major pair slot): written by the sparse backward */
        bin_tile_keys,          /* direct binning: the preprocess leaves every pair's key in its tile's row -- [T][1024] on images of many tiles
                                 * (SG_FLAG_SHORT_LISTS), [T][16384] on images of few tiles (SG_FLAG_LONG_ROWS) */
        bin_bytes;
    /* image workspace */
    size_t img_final_T, img_n_contrib, img_ckpt, img_bytes;
    /* backward workspace */
    size_t bwd_bytes;
} SgLayout;

/* SG_ABI_VERSION changes whenever a struct of this header changes layout or an entry point changes its signature or the meaning of
 * a workspace size (round 4 inserted SgLayout.bin_rec_valid and enlarged the K-frame bwd_ws without a signal: ADVICE r4).  A C
 * caller checks `sg_abi_version() == SG_ABI_VERSION` once after dlopen; the Python host does (sings_amd/_lib.py).  New SgLayout
 * fields are appended from now on.  Workspaces are sized ONLY through sg_layout (one frame) / sg_frames_layout (K frames) -- never
 * as a multiple computed by the caller.
 * History: 5 = round 5's first tree; 6 = SgTriplane gained `feature_minor` (+ `reserved`) at its end, sg_weight_grad_ws_bytes grew
 * (two partial slabs per workgroup for <= 64 outputs); entry points added since 5: sg_linear_backward_fan, sg_rows_laplacian,
 * sg_scales_head_forward / _backward. */
#define SG_ABI_VERSION 7
int sg_abi_version(void);
const char *sg_version(void);
const char *sg_last_error(void);

/* SG_FLAG_SHORT_LISTS: the caller asserts that no tile's list exceeds 1024 entries -- what a compositing workgroup sorts
 * itself -- (known from an earlier forward of the same scene: a pre-sized engine).  The two kernels that sort longer lists are then not launched at all -- such lists are
 * sorted by the compositing workgroups themselves -- which saves their launch latency (3.6 us of a 350-us cfg3 view).  If a
 * longer list does turn up the forward does NOT follow it: it renders the background, writes no gradients, and
 * sg_read_num_rendered / num_rendered_host report SG_NUM_RENDERED_LONG_LIST (re-run without the flag).
 * On images of many tiles (no per-workgroup tile histogram: more than 4096 tile counters) the flag also selects DIRECT binning: the
 * preprocess writes a pair's key (depth bits << 32 | id) straight into row `tile` of a [T][1024] array at the rank its counting
 * atomic returned -- no (Gaussian, tile, rank) records, no scatter pass over the pairs; the compositing workgroup sorts its row.
 * Lists, ranges, images and gradients are the same bits either way. */
#define SG_FLAG_SHORT_LISTS 1
#define SG_COUNT_FLAG_HALF_ROWS 4u   /* in the count_signal word: see SgRasterSettings */
#define SG_COUNT_FLAG_HALF_LONG_ROWS 8u   /* ... no tile list longer than 8192 entries: SG_FLAG_LONG_ROWS is safe for a similar frame */
/* SG_FLAG_WS_CLEAN: the caller vouches that the counters at the head of `binning_ws` (the first sg_layout().bin_ranges
 * bytes) are zero: the workspace was zero-filled there after allocation, or its last use was a forward of this library that
 * returned 0 -- every forward leaves them zeroed (its last kernel clears what the next forward's first kernel counts into).
 * The per-call zeroing launch (~7 us of a 350-us cfg3 view, launch gap included) is then skipped.  Without the flag
 * the workspace may hold anything. */
#define SG_FLAG_WS_CLEAN 2
/* SG_FLAG_THROUGHPUT: the caller keeps SEVERAL views in flight on the device (sings_amd.engine.ViewBatch over HIP streams).  The
 * library then does not spend extra instructions on latency: on frames of few tiles (<= 4096: an avatar) it otherwise
 * composites every tile of more than 1024 entries with four workgroups, one per quadrant -- 15 % off the forward composite of a
 * frame rendered alone, ~3 % slower when other views already fill the idle SIMDs.  Must be the same in the forward and the
 * backward call of a view (it decides the layout of the per-entry quadrant masks). */
#define SG_FLAG_THROUGHPUT 4
/* The forward in two calls on the SAME workspaces, for callers that schedule the halves on different streams (ordered by their own
 * events): SG_FLAG_FORWARD_BINNING runs the per-Gaussian kernel and the tile binning only (latency-bound, light on the vector ALUs),
 * SG_FLAG_FORWARD_COMPOSITE the per-tile composite only (VALU-bound).  bench.py's avatar step runs the composite kernels of one
 * batch of frames on one stream while the binning / loss / per-Gaussian backward of the next batch run beside them on another.
 * Neither flag: the whole forward.  num_rendered_host is honoured by the call that runs the binning. */
#define SG_FLAG_FORWARD_BINNING 8
#define SG_FLAG_FORWARD_COMPOSITE 16
/* SG_FLAG_SH_PLANAR (backward calls): dL_dsh is COEFFICIENT-major, [M][P][3] instead of the reference's [P][M][3], and only the
 * (sh_degree + 1)^2 coefficient planes in use are written (or, with accumulate, added to); the others are never touched -- zero them
 * once.  The reference always allocates M = 16 rows (sings/rec/models/modules/decoders.py:34) and trains at sh_degree 0
 * (human_complex.yaml:34): in the [P][16][3] layout 45 of the 55 gradient floats per Gaussian are structural zeros that every
 * frame writes, every fold reads and every all-reduce carries.  With the planes in use at the head of the SH block the
 * canonical-Gaussian gradient of a step is ONE contiguous prefix of the flat buffer (sings_amd.engine: 10 floats per Gaussian on
 * the avatar) and the frame-parallel collective moves 6 MB instead of 33. */
#define SG_FLAG_SH_PLANAR 32
/* SG_FLAG_LONG_ROWS: images of FEW tiles (at most 4096 tile counters: the per-workgroup tile histogram regime, an avatar frame): the caller
 * asserts that no tile's list exceeds 16384 entries -- a row of `bin_tile_keys` there -- and gets direct binning as under
 * SG_FLAG_SHORT_LISTS (the preprocess writes every pair's key into its tile's row at its final rank; no pair scatter pass).  Lists of more
 * than 1024 entries are still sorted by the long-list kernels, from their rows.  A longer list: nothing is composited,
 * SG_NUM_RENDERED_LONG_LIST, as for SG_FLAG_SHORT_LISTS.  Ignored on images of many tiles.  Same bits either way. */
#define SG_FLAG_LONG_ROWS 64
#define SG_NUM_RENDERED_LONG_LIST (-2)
/* Workspace sizing.  capacity_pairs = upper bound on R = sum of tiles touched. */
int sg_layout(int P, int width, int height, size_t capacity_pairs, SgLayout *out);

/* Forward: replaces _C.rasterize_gaussians.  Exactly one of (shs | colors_precomp) and one
 * of (scales+rotations | cov3D_precomp) is non-NULL, as GaussianRasterizer.forward enforces.
 * out_color [3,H,W], radii [P] are fully written.  If num_rendered_host != NULL the call
 * stores R there before it returns (R > capacity_pairs => results invalid, retry with a larger
 * workspace; status stays 0): by synchronising the stream, or -- with SgRasterSettings.count_signal
 * set -- as soon as the binning kernel has published the count, the rest of the forward still running.  point_keys_out: optional [capacity] u64
 * receiving the upstream-format sorted keys (tile << 32 | depth bits). */
int sg_rasterize_forward(const SgRasterSettings *s, int P, const float *means3D, const float *shs,
                         const float *colors_precomp, const float *opacities, const float *scales,
                         const float *rotations, const float *cov3D_precomp, void *geom_ws,
                         void *binning_ws, size_t capacity_pairs, void *image_ws, float *out_color,
                         int32_t *radii, int write_point_keys, int64_t *num_rendered_host,
                         void *stream);

/* Backward: replaces _C.rasterize_gaussians_backward.  All gradient outputs are fully
 * written (no pre-zeroing needed).  Pointers for absent inputs may be NULL. */
int sg_rasterize_backward(const SgRasterSettings *s, int P, const float *means3D, const float *shs,
                          const float *colors_precomp, const float *opacities, const float *scales,
                          const float *rotations, const float *cov3D_precomp, const int32_t *radii,
                          const void *geom_ws, const void *binning_ws, size_t capacity_pairs,
                          const void *image_ws, void *bwd_ws, const float *dL_dout_color,
                          float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dsh, float *dL_dcolors,
                          float *dL_dopacity, float *dL_dscales, float *dL_drotations,
                          float *dL_dcov3D, void *stream);

/* The two halves of sg_rasterize_backward as separate calls, for callers that render several views of the SAME Gaussians per
 * optimisation step (frame-parallel training: every rank renders a batch of frames, gs_trainer.py:207-215 is one frame per step):
 *   _records   : the per-tile composite backward of one view -> one gradient record per (tile, Gaussian) pair in bwd_ws;
 *   _gaussians : per Gaussian, the sum of its records and the chain rule to the inputs.  accumulate = 0: the gradient outputs are
 *                written; != 0: the view's gradients are ADDED to what the outputs hold (dL_dmeans2D, the densifier's per-view
 *                statistic, is always written).  The views of a step can therefore share ONE gradient buffer -- the first view
 *                writes, the others add, in an order the caller fixes with stream events (deterministic) -- instead of one
 *                buffer per view and a pass that sums them. */
int sg_rasterize_backward_records(const SgRasterSettings *s, int P, const void *geom_ws, const void *binning_ws,
                                  size_t capacity_pairs, const void *image_ws, void *bwd_ws, const float *dL_dout_color,
                                  void *stream);
int sg_rasterize_backward_gaussians(const SgRasterSettings *s, int P, const float *means3D, const float *shs,
                                    const float *colors_precomp, const float *opacities, const float *scales,
                                    const float *rotations, const float *cov3D_precomp, const int32_t *radii,
                                    const void *geom_ws, const void *binning_ws, size_t capacity_pairs, const void *bwd_ws,
                                    int accumulate, float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dsh, float *dL_dcolors,
                                    float *dL_dopacity, float *dL_dscales, float *dL_drotations, float *dL_dcov3D, void *stream);

/* markVisible of the upstream module: present[i] = view-space z > 0.2 */
int sg_mark_visible(int P, const float *means3D, const float *viewmatrix, const float *projmatrix,
                    uint8_t *present, void *stream);

/* Reads R written by the last forward into this binning workspace (synchronises); SG_NUM_RENDERED_LONG_LIST if that
 * forward ran with SG_FLAG_SHORT_LISTS and met a longer list. */
int sg_read_num_rendered(const void *binning_ws, int64_t *num_rendered_host, void *stream);

/* `slots` 64-bit words of pinned, mapped, coherent host memory for SgRasterSettings.count_signal: *host_out is the address the
 * host polls, *device_out the address kernels store to (the same memory).  Caller-owned until sg_signal_free(*host_out). */
int sg_signal_alloc(int slots, void **host_out, void **device_out);
int sg_signal_free(void *host);

/* ---- LBS-fused path: canonical Gaussians + joint transforms in, image out ------------------
 * Replaces the LBS block of SinGS.forward (sings/rec/models/sings_hybrid.py:398-428; lbs_extra at
 * sings/rec/utils/body_model/lbs.py:59-74; matrix_to_quaternion at
 * sings/rec/utils/geometry/rotations.py:98-149) AND the rasterizer call that follows it
 * (gs_renderer_single.py:87-95): posed means / quaternions / T[N,4,4] stay in registers. */
typedef struct SgSkinInputs {
    int32_t J;                /* joints: 24 (SMPL) or 52 (SMPL-H); <= 64 */
    int32_t rot_format;       /* SG_ROT_CANON_MATRIX (0) or SG_ROT_CANON_6D (1): layout of rot_canon */
    const float *xyz_canon;   /* [P,3] canonical means */
    const float *rot_canon;   /* NULL = identity (isotropic, sings_hybrid.py:358-361); [P,9] row-major rotation matrices; or
                                 [P,6] in the decoder's 6-D form -- rotation_6d_to_matrix (sings_hybrid.py:356-357,
                                 rotations.py:545-566) is then evaluated inside the kernel and dL_drot_canon is [P,6] */
    const float *lbs_weights; /* [P,J] skinning weights */
    const float *A;           /* [J,16] row-major cano->pose joint transforms (A_t2pose @ inv_A_t2cano) */
    const float *smpl_scale;  /* [1] or NULL */
    const float *transl;      /* [3] or NULL */
    const float *ext_trans;   /* [3]  \                                                    */
    const float *ext_rot;     /* [9]   > ext_tfs of sings_hybrid.py:421-428, all or none;   */
    const float *ext_scale;   /* [1]  /  forward only (the reference uses them under no_grad) */
} SgSkinInputs;
enum { SG_ROT_CANON_MATRIX = 0, SG_ROT_CANON_6D = 1 };

/* floats the caller must provide as `skin_ws` to sg_skinned_backward (and as `ws` to sg_lbs_backward) */
size_t sg_skin_ws_floats(int P);

/* Stand-alone lbs_extra (sings/rec/utils/body_model/lbs.py:59-74, call sites sings_hybrid.py:400-406, :526-533), for
 * callers that keep SinGS.forward unchanged: T [P,16] = lbs_weights [P,J] . A [J,16] (row-major 4x4 per point; may be NULL)
 * and verts [P,3] = (T [v;1])[:3].  Backward: upstream gradients dT [P,16] and / or dverts [P,3] (either may be NULL)
 * -> dv [P,3] (may be NULL) and dA [J,16]; lbs_weights get no gradient (the reference detaches them, sings_hybrid.py:724).
 * 1 <= J <= 64.  Deterministic (fixed-order reduction of dA, no atomics). */
int sg_lbs_forward(int P, int J, const float *lbs_weights, const float *A, const float *v, float *T_out, float *verts_out,
                   void *stream);
int sg_lbs_backward(int P, int J, const float *lbs_weights, const float *A, const float *v, const float *dT,
                    const float *dverts, float *ws, float *dv, float *dA, void *stream);

/* SMPL(-H) kinematic chain (sings/rec/utils/body_model/smpl.py:415-513: batch_rodrigues + batch_rigid_transform, what
 * SMPL.forward / SMPLH.forward produce as `A`, smpl_layer.py:492-600): pose [B,J,3] axis-angle, joints_rest [J,3],
 * parents [J] (parents[0] < 0, parents[i] < i) -> A [B,J,16] = G - pad(G [J;0]), optionally right-multiplied per joint by
 * post [J,16] (inv(A_t2cano) of sings_hybrid.py:398-399; may be NULL).  One launch instead of ~100 (forward) / ~300
 * (autograd).  Backward: dA [B,J,16] -> dpose [B,J,3] and, if djoints != NULL, the per-frame djoints [B,J,3] (the caller
 * sums over B).  1 <= J <= 64. */
int sg_joint_transforms(int B, int J, const float *pose, const float *joints_rest, const int32_t *parents, const float *post,
                        float *A_out, void *stream);
int sg_joint_transforms_backward(int B, int J, const float *pose, const float *joints_rest, const int32_t *parents,
                                 const float *post, const float *dA, float *dpose, float *djoints, void *stream);

/* Stand-alone matrix_to_quaternion (sings/rec/utils/geometry/rotations.py:98-149; call site sings_hybrid.py:419):
 * matrices [N,9] row-major -> quaternions [N,4] (real first), bit-identical to the reference expression evaluated by
 * torch on the same GPU; backward
 * dq [N,4] -> dmatrices [N,9] (autograd of that expression: only the selected candidate receives gradient). */
int sg_matrix_to_quaternion(int N, const float *matrices, float *quaternions, void *stream);
int sg_matrix_to_quaternion_backward(int N, const float *matrices, const float *dq, float *dmatrices, void *stream);

/* The other conversions of sings/rec/utils/geometry/rotations.py on the path, one launch each way (the reference
 * runs each as 10-25 element-wise torch kernels).  `op`:
 *   SG_ROT_QUATERNION_TO_MATRIX      :38-66    in [N,4] (real first, any norm)  -> out [N,9] row-major
 *   SG_ROT_6D_TO_MATRIX              :545-566  in [N,6] (a1, a2)                 -> out [N,9], rows b1, b2, b1 x b2
 *                                              (every Gaussian, sings_hybrid.py:356-357)
 *   SG_ROT_AXIS_ANGLE_TO_QUATERNION  :482-511  in [N,3]                          -> out [N,4]
 *   SG_ROT_QUATERNION_TO_AXIS_ANGLE  :514-545  in [N,4]                          -> out [N,3]
 * sg_rotation_convert: value.  sg_rotation_convert_backward: d_out [N, out width] -> d_in [N, in width], the gradient
 * torch.autograd gives for the reference expression.  sg_quaternion_multiply: standardize(a * b) (:393-407 with :357,
 * :372-390), a, b, out [N,4]; its backward writes da and db. */
enum { SG_ROT_QUATERNION_TO_MATRIX = 0, SG_ROT_6D_TO_MATRIX = 1, SG_ROT_AXIS_ANGLE_TO_QUATERNION = 2,
       SG_ROT_QUATERNION_TO_AXIS_ANGLE = 3 };
int sg_rotation_convert(int op, int N, const float *in, float *out, void *stream);
int sg_rotation_convert_backward(int op, int N, const float *in, const float *d_out, float *d_in, void *stream);
int sg_quaternion_multiply(int N, const float *a, const float *b, float *out, void *stream);
int sg_quaternion_multiply_backward(int N, const float *a, const float *b, const float *d_out, float *da, float *db,
                                    void *stream);

/* Forward.  `scales` are the CANONICAL scales [P,3]; optional outputs posed_xyz [P,3],
 * posed_rotq [P,4] (real first, not normalised), posed_scales [P,3] may be NULL. */
int sg_skinned_forward(const SgRasterSettings *s, int P, const SgSkinInputs *skin, const float *shs,
                       const float *opacities, const float *scales, void *geom_ws, void *binning_ws,
                       size_t capacity_pairs, void *image_ws, float *out_color, int32_t *radii,
                       float *posed_xyz, float *posed_rotq, float *posed_scales,
                       int64_t *num_rendered_host, void *stream);

/* Backward (LBS^T).  dL_dposed_xyz_in / dL_dposed_rotq_in: optional upstream gradients on the posed
 * outputs.  Outputs fully written: dL_dxyz_canon [P,3], dL_drot_canon [P,9] (may be NULL),
 * dL_dscales [P,3], dL_dopacity [P], dL_dsh [P,M,3], dL_dmeans2D [P,3], dL_dA [J,16],
 * dL_dtransl [3] (may be NULL).  smpl_scale, lbs_weights and ext_tfs receive no gradient (they are
 * data / buffers in the reference: sings_hybrid.py:724, gs_trainer.py:230). */
int sg_skinned_backward(const SgRasterSettings *s, int P, const SgSkinInputs *skin, const float *shs,
                        const float *opacities, const float *scales, const int32_t *radii,
                        const void *geom_ws, const void *binning_ws, size_t capacity_pairs,
                        const void *image_ws, void *bwd_ws, float *skin_ws, const float *dL_dout_color,
                        const float *dL_dposed_xyz_in, const float *dL_dposed_rotq_in,
                        float *dL_dxyz_canon, float *dL_drot_canon, float *dL_dscales,
                        float *dL_dopacity, float *dL_dsh, float *dL_dmeans2D, float *dL_dA,
                        float *dL_dtransl, void *stream);

/* Second half of sg_skinned_backward alone (the first half is sg_rasterize_backward_records): accumulate != 0 adds the frame's
 * canonical-Gaussian gradients (dL_dxyz_canon, dL_drot_canon, dL_dscales, dL_dopacity, dL_dsh) to the outputs; dL_dmeans2D, dL_dA
 * and dL_dtransl are per frame and always written. */
int sg_skinned_backward_gaussians(const SgRasterSettings *s, int P, const SgSkinInputs *skin, const float *shs,
                                  const float *opacities, const float *scales, const int32_t *radii, const void *geom_ws,
                                  const void *binning_ws, size_t capacity_pairs, const void *bwd_ws, float *skin_ws,
                                  int accumulate, const float *dL_dposed_xyz_in, const float *dL_dposed_rotq_in,
                                  float *dL_dxyz_canon, float *dL_drot_canon, float *dL_dscales, float *dL_dopacity,
                                  float *dL_dsh, float *dL_dmeans2D, float *dL_dA, float *dL_dtransl, void *stream);

/* ---- K frames of the SAME Gaussians per call (round 4) ------------------------------------------------------------------------
 * The reference hands the model 16 frames per call (SinGS.forward_chunk, sings/rec/models/sings_hybrid.py:474-569, consumed one frame
 * at a time by the render loop at sings/rec/trainer/gs_trainer.py:684-714), and a frame-parallel training step renders K frames of the
 * same canonical Gaussians per rank.  The *_frames entry points take those K frames in ONE call and ONE dispatch per kernel: the
 * latency-bound kernels of a frame (binning, long-list sort, loss, reductions) fill the chip with K frames' worth of workgroups, the
 * canonical inputs are read from HBM once for the K poses / cameras, and the per-Gaussian backward sums the K frames' gradients in
 * registers and writes the gradient row once -- in frame order, bit for bit what K single-frame calls leave behind when the first
 * writes the gradient buffer and the others add to it (accumulate = 1).  K = 1 IS the single-frame call (same kernels).
 *
 * Layout of a K-frame call: every per-frame array is K consecutive single-frame arrays --
 *   workspaces : geom_ws / binning_ws / image_ws / bwd_ws = K x sg_layout(...).{geom,bin,img,bwd}_bytes (sg_frames_layout;
 *                bwd_ws, K > 1: the K record buffers + the aligned record-sum scratch of sg_rasterize_backward_gaussians_frames
 *                behind them: take the size from sg_frames_layout, it is NOT K x bwd_bytes),
 *   out_color / dL_dout_color [K,3,H,W], radii [K,P], dL_dmeans2D [K,P,3], posed_* [K,P,.], dL_dA [K,J,16], dL_dtransl [K,3],
 *   skin->A [K,J,16]; skin->transl [K,3] (transl_stride 3) or [3] shared (0);
 *   cameras: camera_stride 1: s->viewmatrix [K,16], s->projmatrix [K,16], s->campos [K,3]; 0: one camera for all frames
 *   (the shipped kit's sequences have ONE static camera).  Image size, fov, sh_degree, bg, scale_modifier are shared. */
#define SG_MAX_FRAMES 16
typedef struct SgFrameBatch {
    int32_t K;               /* frames in this call, 1..SG_MAX_FRAMES */
    int32_t camera_stride;   /* 0 | 1 */
    int32_t transl_stride;   /* 0 | 3 (floats) */
    int32_t reserved;        /* 0 */
} SgFrameBatch;
/* sizes of the K-frame workspaces (K x the single-frame sizes, which are returned in *one; bwd: see above); any output pointer may
 * be NULL */
int sg_frames_layout(int P, int width, int height, size_t capacity_pairs, int K, SgLayout *one, size_t *geom_bytes,
                     size_t *binning_bytes, size_t *image_bytes, size_t *bwd_bytes);
/* num_rendered_host: NULL, or [K] -- filled by one strided copy + stream synchronisation */
int sg_rasterize_forward_frames(const SgRasterSettings *s, const SgFrameBatch *fb, int P, const float *means3D, const float *shs,
                                const float *colors_precomp, const float *opacities, const float *scales, const float *rotations,
                                const float *cov3D_precomp, void *geom_ws, void *binning_ws, size_t capacity_pairs, void *image_ws,
                                float *out_color, int32_t *radii, int64_t *num_rendered_host, void *stream);
int sg_skinned_forward_frames(const SgRasterSettings *s, const SgFrameBatch *fb, int P, const SgSkinInputs *skin, const float *shs,
                              const float *opacities, const float *scales, void *geom_ws, void *binning_ws, size_t capacity_pairs,
                              void *image_ws, float *out_color, int32_t *radii, float *posed_xyz, float *posed_rotq,
                              float *posed_scales, int64_t *num_rendered_host, void *stream);
int sg_read_num_rendered_frames(const void *binning_ws, int P, int width, int height, size_t capacity_pairs, int K,
                                int64_t *num_rendered_host, void *stream);
int sg_rasterize_backward_records_frames(const SgRasterSettings *s, const SgFrameBatch *fb, int P, const void *geom_ws,
                                         const void *binning_ws, size_t capacity_pairs, const void *image_ws, void *bwd_ws,
                                         const float *dL_dout_color, void *stream);
/* accumulate != 0: the K frames' sum is ADDED to the gradient outputs (several calls of one optimisation step, one buffer) */
int sg_rasterize_backward_gaussians_frames(const SgRasterSettings *s, const SgFrameBatch *fb, int P, const float *means3D,
                                           const float *shs, const float *colors_precomp, const float *opacities,
                                           const float *scales, const float *rotations, const float *cov3D_precomp,
                                           const int32_t *radii, const void *geom_ws, const void *binning_ws,
                                           size_t capacity_pairs, void *bwd_ws /* records in; K > 1: its tail is scratch */,
                                           int accumulate, float *dL_dmeans3D,
                                           float *dL_dmeans2D, float *dL_dsh, float *dL_dcolors, float *dL_dopacity,
                                           float *dL_dscales, float *dL_drotations, float *dL_dcov3D, void *stream);
size_t sg_skin_ws_floats_frames(int P, int K);
int sg_skinned_backward_gaussians_frames(const SgRasterSettings *s, const SgFrameBatch *fb, int P, const SgSkinInputs *skin,
                                         const float *shs, const float *opacities, const float *scales, const int32_t *radii,
                                         const void *geom_ws, const void *binning_ws, size_t capacity_pairs, const void *bwd_ws,
                                         float *skin_ws, int accumulate, const float *dL_dposed_xyz_in,
                                         const float *dL_dposed_rotq_in, float *dL_dxyz_canon, float *dL_drot_canon,
                                         float *dL_dscales, float *dL_dopacity, float *dL_dsh, float *dL_dmeans2D, float *dL_dA,
                                         float *dL_dtransl, void *stream);
/* K photometric losses + gradients in three launches (HumanSceneLoss.forward of K frames, losses/loss.py:55-69): raw, dL_draw,
 * pred_out, gt_out [K,3,H,W]; gt_rgb at + gt_stride floats per frame, mask at + mask_stride floats per frame (0: one target /
 * mask for all frames); `ws`: K x sg_photo_loss_ws_bytes(W, H); losses [K,4]; upstream [2] (shared) or NULL */
int sg_photo_loss_frames(int K, int width, int height, float l1_w, float ssim_w, const float *raw, const float *gt_rgb,
                         size_t gt_stride, const float *mask, size_t mask_stride, const float *bg, void *ws, float *pred_out,
                         float *gt_out, float *losses, const float *upstream, float *dL_draw, void *stream);
/* The gradients alone, K frames in one launch, over the workspace a forward-only sg_photo_loss_frames call left behind (the K-frame
 * form of sg_photo_loss_backward below): `upstream` NULL = (1, 1) for every frame, upstream_stride 0 = one pair [2] for all
 * frames, 2 = a pair per frame [K,2] (device memory) */
int sg_photo_loss_backward_frames(int K, int width, int height, float l1_w, float ssim_w, const float *raw, const float *gt_rgb,
                                  size_t gt_stride, const float *mask, size_t mask_stride, const float *bg, const void *ws,
                                  const float *upstream, int upstream_stride, float *dL_draw, void *stream);

/* ---- photometric loss of one view, forward + gradient -------------------------------------------
 * Replaces, for the L1 and SSIM terms of HumanSceneLoss.forward (sings/rec/losses/loss.py:55-69):
 *   pred = clamp(raw, 0, 1)                         gs_renderer_single.py:96
 *   gt   = gt_rgb * mask + bg * (1 - mask)          loss.py:58
 *   losses[0] = l1_w   * |pred - gt|.sum() / mask.sum()                 losses/utils.py:16-20, loss.py:61-63
 *   losses[1] = ssim_w * (1 - ssim(pred, gt)) * mask.sum() / (H * W)    losses/utils.py:28-70, loss.py:65-69
 *   losses[2] = the unweighted L1, losses[3] = mean SSIM
 * and the autograd backward of losses[0] * upstream[0] + losses[1] * upstream[1] down to the rasterizer
 * output: dL_draw [3,H,W] (upstream NULL = (1, 1); dL_draw NULL = forward only).  All pointers are device
 * memory; raw / gt_rgb [3,H,W], mask [H,W], bg [3], losses [4]; pred_out / gt_out [3,H,W] optional (the
 * reference's extras_dict 'pred_img' / 'gt_img').  `ws`: sg_photo_loss_ws_bytes(W, H) bytes of scratch.
 * The LPIPS term of the reference is a VGG forward pass and is not part of this library. */
size_t sg_photo_loss_ws_bytes(int width, int height);
int sg_photo_loss(int width, int height, float l1_w, float ssim_w, const float *raw, const float *gt_rgb,
                  const float *mask, const float *bg, void *ws, float *pred_out, float *gt_out, float *losses,
                  const float *upstream, float *dL_draw, void *stream);
/* The gradient alone, for an autograd backward whose upstream weights are only known later (loss.backward() of
 * sum(loss_dict.values()), gs_trainer.py:240-262): `ws` is the workspace a forward-only sg_photo_loss call on the SAME
 * inputs left behind (the mask's partial sums + loss scalars; not modified), `upstream` [2] device memory (NULL = (1, 1)).  Since round 6 a call
 * WITH `dL_draw` costs one march over the image (forward and gradient are one kernel): a host that will differentiate should ask for
 * the unit-upstream gradient in the forward call and scale it (sings_amd/photo_loss.py does).
 * Forward + this call cost what the fused call costs, and nothing has to be read back to compare the two weights. */
int sg_photo_loss_backward(int width, int height, float l1_w, float ssim_w, const float *raw, const float *gt_rgb,
                           const float *mask, const float *bg, const void *ws, const float *upstream, float *dL_draw,
                           void *stream);

/* ---- geometry-preserving regularisers, value + gradient ---------------------------------
 * Replace sings/rec/losses/loss_items.py (called from gs_trainer.py:355-399).  All pointers device memory;
 * `loss` [1]; `upstream` [1] = d(total)/d(this loss), NULL = 1; gradient outputs may be NULL (value only).
 * `ws`: sg_reg_ws_bytes(rows) bytes (sg_knn_ws_bytes(N) for the edge loss).  Deterministic (no float atomics). */
size_t sg_reg_ws_bytes(int rows);
/* RegionLaplacianLoss_v2.forward / forward_hands (:93-192) on the static graph: CSR (row_ptr [V+1], col [nnz], both
 * directions of every edge whose endpoints share a label), deg_inv [V] = 1/deg (0 if isolated),
 * vscale [V] = weight(region) / (V_region * C) (0 outside the summed regions); x [V,C]; g_ws [V,C] scratch.
 *   loss = sum_i vscale_i * |(D^-1 A x - x)_i|^2 */
int sg_region_laplacian(int V, int C, const float *x, const int *row_ptr, const int *col, const float *deg_inv,
                        const float *vscale, void *ws, float *g_ws, float *loss, const float *upstream,
                        float *dL_dx, void *stream);
/* RegionLaplacianLoss_v2 with laplacian_type = "cotangent" (:150-165, :183-192): the regions overlap (every face that touches a
 * vertex of the region), so the operator is a stack of R weighted rows -- one per (region, vertex of the region) -- over the global
 * vertex array x [V,C]: CSR (row_ptr [R+1], col [nnz] = global vertex, val [nnz] = cot_laplacian's off-diagonal weight; no diagonal,
 * as in the reference's product), rscale [R] = weight(region) / (rows of the region * C); the transposed CSR (t_row_ptr [V+1],
 * t_row [nnz] = stacked row, t_val [nnz]) is needed for dL_dx only.  g_ws [R,C] scratch, ws: sg_reg_ws_bytes(R).
 *   loss = sum_r rscale_r |sum_e val_e x[col_e]|^2 */
int sg_rows_laplacian(int R, int V, int C, const float *x, const int *row_ptr, const int *col, const float *val,
                      const float *rscale, const int *t_row_ptr, const int *t_row, const float *t_val, void *ws, float *g_ws,
                      float *loss, const float *upstream, float *dL_dx, void *stream);
/* pytorch3d.loss.mesh_edge_loss(mesh, target_length=0) (gs_trainer.py:366): (1/E) sum_e |v0 - v1|^2; CSR with both
 * directions of the E unique edges */
int sg_mesh_edge_loss(int V, int E, const float *x, const int *row_ptr, const int *col, void *ws, float *loss,
                      const float *upstream, float *dL_dx, void *stream);
/* L2Norm.forward (:15-54): lambdas = (lambda_xyz_offsets, lambda_scales_diff, lambda_max_scale, max_scale_threshold,
 * lambda_min_opacity, min_opacity_threshold); xyz_offsets [N,3], scales [N,3] (column 0 is used), opacity [N,1];
 * any of the three inputs may be NULL (= not in norm_list) */
int sg_l2norm_reg(int N, const float *xyz_offsets, const float *scales, const float *opacity, const float *lambdas6,
                  void *ws, float *loss, const float *upstream, float *d_offsets, float *d_scales, float *d_opacity,
                  void *stream);
/* GaussiansEdgeLoss.forward (:57-90): mean_edge_i = mean Euclidean distance to the K-1 nearest other points
 * (K = 9 in the reference, K in {5, 9, 17} supported, N >= K), exact, uniform-grid search instead of the brute-force
 * pytorch3d.ops.knn_points; loss = mean((scales[:,0] - mean_edge)^2), gradient to scales only (edge lengths are
 * detached in the reference).  mean_edge_out [N] optional; scales NULL = neighbour search only. */
size_t sg_knn_ws_bytes(int N);
int sg_gaussian_edge_loss(int N, int K, const float *xyz, const float *scales, void *ws, float *mean_edge_out,
                          float *loss, const float *upstream, float *d_scales, void *stream);
/* The same in two calls on the same workspace, for callers that schedule the halves themselves: `prepare` builds the two
 * search grids over xyz (fifteen small, latency-bound launches), `finish` runs the neighbour query -- one kernel that fills
 * the GPU for ~0.25 ms at 150 k points -- and the loss.  Nothing else may use `ws` in between; the two calls may be on
 * different streams if the caller orders them (event).  sg_gaussian_edge_loss == prepare + finish on one stream. */
int sg_gaussian_edge_prepare(int N, const float *xyz, void *ws, void *stream);
int sg_gaussian_edge_finish(int N, int K, const float *scales, void *ws, float *mean_edge_out, float *loss,
                            const float *upstream, float *d_scales, void *stream);

/* ---- attribute decode ---------------------------------------------------------------------
 * Multi-resolution tri-plane features: HexPlaneField.forward (sings/rec/models/modules/hexplane.py:46-105,163-190):
 * per scale s the product over the planes (x,y), (x,z), (y,z) of F.grid_sample(bilinear, border, align_corners=True),
 * concatenated over scales.  planes[s][c]: the reference parameter grids[s][c] of shape [1, feat, res[s][b], res[s][a]]
 * for the coordinate pair (a, b) of plane c; feat must be 32; aabb = HexPlaneField.aabb (row 0, row 1).
 * forward: feats [N, n_scales * feat].  backward: dplanes[s][c] in the same layout (fully written; NULL = skip),
 * dxyz [N,3] optional.  `ws`: sg_triplane_ws_bytes() bytes for the forward, sg_triplane_bwd_ws_bytes(tp, N) for the
 * backward (both calls re-transpose the current parameters).  The backward is a gather: the points are counting-sorted
 * by texel cell once per projection and every texel block sums its points' rows in LDS -- no global float atomics;
 * the order of the additions inside a texel is not fixed (reproducible to rounding, like grid_sample's backward). */
typedef struct {
    int n_scales, feat;
    int res[4][3];
    const float *planes[4][3];
    float aabb[2][3];
    /* ABI 6: 1 = every plane (and, in the backward, every dplanes[s][c]) is FEATURE-MINOR in memory -- [H][W][feat], i.e. a
     * [1, feat, H, W] tensor in torch's channels_last memory format -- and is used IN PLACE: no texel-major copy of the parameters
     * per call (18 us forward, 28 us in the backward's preparation at the training size), no copy of the gradients back (23 us);
     * the caller zero-fills dplanes before the backward.  0: the reference's [feat][H][W] tensors, as before. */
    int feature_minor;
    int reserved;
} SgTriplane;
size_t sg_triplane_ws_bytes(const SgTriplane *tp);
size_t sg_triplane_bwd_ws_bytes(const SgTriplane *tp, int N);
int sg_triplane_forward(const SgTriplane *tp, int N, const float *xyz, void *ws, float *feats, void *stream);
int sg_triplane_backward(const SgTriplane *tp, int N, const float *xyz, void *ws, const float *dfeats,
                         float *const dplanes[4][3], float *dxyz, void *stream);
/* The same in two calls on ONE workspace: `_prepare` (texel-major planes, zeroed gradient planes, the three counting sorts of the
 * points) needs only the points and the planes -- a quarter of the backward, all latency -- and may run any time after the forward,
 * e.g. on a side stream while the decoders work; `_prepared` then needs dL/dfeats.  The planes must not change in between. */
int sg_triplane_backward_prepare(const SgTriplane *tp, int N, const float *xyz, void *ws, void *stream);
int sg_triplane_backward_prepared(const SgTriplane *tp, int N, const float *xyz, void *ws, const float *dfeats,
                                  float *const dplanes[4][3], float *dxyz, void *stream);
/* Bias + activation around the decoders' library GEMMs (modules/decoders.py:16-110).  act: 0 identity, 1 GELU (erf),
 * 2 sigmoid(z + row_offset[n]) (AppearanceDecoder.opacity_offset; row_offset may be NULL), 3 log(exp(z) + 1).
 * forward: z = y + bias (stored if z_out != NULL), h = act(z); y, z_out, h_out [N,C] (h_out may alias y).
 * backward: dz = dh * act'(z), dbias [C] = column sums of dz (fixed order, deterministic); C <= 128;
 * `ws`: sg_bias_act_ws_bytes(N, C). */
size_t sg_bias_act_ws_bytes(int N, int C);
int sg_bias_act_forward(int N, int C, int act, const float *y, const float *bias, const float *row_offset,
                        float *z_out, float *h_out, void *stream);
int sg_bias_act_backward(int N, int C, int act, const float *z, const float *row_offset, const float *dh, void *ws,
                         float *dz, float *dbias, void *stream);

/* The isotropic scale head's tail (modules/decoders.py:88-94, the `.repeat(1, 3)` of :96-98): z [N,1] -> scales [N,3] = log(exp(z) + 1)
 * and scales_aux [N,3] = z, every row three times; backward dz [N,1] = (sum_c dscales) e^z / (e^z + 1) + sum_c daux (either may be
 * NULL).  One launch each way. */
int sg_scales_head_forward(int N, const float *z, float *scales_out, float *aux_out, void *stream);
int sg_scales_head_backward(int N, const float *z, const float *dscales, const float *daux, float *dz, void *stream);

/* One decoder layer (nn.Linear + activation, modules/decoders.py:41-49, 75-94) as ONE kernel each way on the matrix cores
 * (v_mfma_f32_32x32x2_f32, fp32): bias and activation in the GEMM's epilogue, the activation derivative in the prologue of
 * the input-gradient GEMM.  x [N,Cin], W [Cout,Cin] (nn.Linear.weight), bias [Cout] | NULL, 1 <= Cin, Cout <= 128; `act`
 * and `row_offset` as for sg_bias_act_*.
 *   forward : h_out [N,Cout] = act(x W^T + bias); aux_out [N,Cout] | NULL receives what the backward needs besides h:
 *             GELU: gelu'(z) itself (the erf is evaluated once, here), softplus: z; nothing for act = 0 and sigmoid (NULL).
 *   backward: aux = the forward's aux_out (GELU, softplus) or its h_out (sigmoid: s' = h (1 - h)); NULL when act = 0.
 *             dz_out [N,Cout] | NULL = dh * act' (input of sg_weight_grad; NULL when act = 0: dz is dh), dx_out [N,Cin] = dz W. */
int sg_linear_forward(int N, int Cin, int Cout, int act, const float *x, const float *W, const float *bias,
                      const float *row_offset, float *aux_out, float *h_out, void *stream);
int sg_linear_backward(int N, int Cin, int Cout, int act, const float *aux, const float *row_offset, const float *dh,
                       const float *W, float *dz_out, float *dx_out, void *stream);
/* The same with dx_out += dz W (Cin a multiple of 32): an activation that feeds several layers -- the decoder trunk and its heads,
 * decoders.py:75-94 -- collects its gradient in ONE array instead of one array per consumer plus autograd's additions. */
int sg_linear_backward_accumulate(int N, int Cin, int Cout, int act, const float *aux, const float *row_offset, const float *dh,
                                  const float *W, float *dz_out, float *dx_out, void *stream);

/* A FAN backward: a wide layer and up to two NARROW heads that read the same input (decoders.py:75-94: scales[0] 128 beside
 * xyz_offsets 3 and rotations 6; :41-49: shs 48 beside opacity 1).  dx_out [N,Cin] = dz W + dz_0 W_0 + dz_1 W_1 in ONE pass: the heads'
 * columns are extra columns of the wide layer's reduction instead of one read-modify-write pass over dx per head.  Cin a multiple
 * of 32, Cout a multiple of 4, side[0].cout + side[1].cout <= 16 (cout = 0: unused); a head's act is 0 or 2 (sigmoid: aux = its
 * forward output h, dz_out receives dh h (1 - h) for its sg_weight_grad; act = 0: dz is dh, dz_out ignored). */
typedef struct SgLinearSide {
    int cout, act;
    const float *aux;                /* [N,cout] | NULL */
    const float *dh;                 /* [N,cout] */
    const float *W;                  /* [cout,Cin] */
    float *dz_out;                   /* [N,cout] | NULL */
} SgLinearSide;
int sg_linear_backward_fan(int N, int Cin, int Cout, int act, const float *aux, const float *dh, const float *W, float *dz_out,
                           float *dx_out, const SgLinearSide side[2], void *stream);

/* Weight / bias gradient of one decoder layer: dW [Cout,Cin] = dz^T x, db [Cout] = column sums of dz (db may be NULL);
 * dz [N,Cout], x [N,Cin]; Cin in {32, 64, 96, 128}, Cout <= 128.  fp32 on the matrix cores, deterministic.
 * `ws`: sg_weight_grad_ws_bytes(N, Cout, Cin). */
size_t sg_weight_grad_ws_bytes(int N, int Cout, int Cin);
int sg_weight_grad(int N, int Cout, int Cin, const float *dz, const float *x, void *ws, float *dW, float *db,
                   void *stream);

/* ---- optional per-kernel timing (bench / profiling only; process-global, not thread-safe).
 * When enabled, every kernel launch of forward/backward is bracketed by hipEvents on the
 * caller's stream.  sg_profile_collect synchronises, adds the elapsed milliseconds and launch
 * counts per kernel id into the caller's arrays (length >= SG_NUM_KERNELS) and drops the events. */
#define SG_NUM_KERNELS 8
enum SgKernelId {
    SG_K_PREPROCESS_FWD = 0, SG_K_PHOTO_LOSS = 1, SG_K_TILE_SCAN = 2, SG_K_TILE_SCATTER = 3,
    SG_K_TILE_SORT = 4, SG_K_RENDER_FWD = 5, SG_K_RENDER_BWD = 6, SG_K_PREPROCESS_BWD = 7
};
int sg_profile_enable(int on);
/* Measurement only: dst[0, bytes) = src[0, bytes) with a plain 16-byte-per-lane copy kernel (both 16-byte aligned, bytes a
 * multiple of 16; four independent loads per lane in flight; non_temporal != 0: non-temporal loads and stores) -- the float4-copy
 * bandwidth SURVEY.md 8(d) names as the denominator of the HBM roofline, measured on the box the benchmark runs on (bench.py:
 * 1 GiB, best of 3 of either form).  Replaces nothing in the reference. */
int sg_copy_probe(void *dst, const void *src, size_t bytes, int non_temporal, void *stream);
int sg_profile_collect(double *total_ms, int64_t *launches, int n);
const char *sg_kernel_name(int id);

#ifdef __cplusplus
}
#endif
#endif
