// Plain (un-skinned) per-Gaussian kernels: the GaussianRasterizer surface.
//
// Replaces preprocessCUDA<3> (+ computeCov2DCUDA in backward) of the rasterizer the reference
// calls at sings/rec/renderer/gs_renderer_single.py:87-95; algorithm per SURVEY.md App. A.1/A.5.
// One lane per Gaussian; 256-thread workgroups; grid = ceil(P/256) >> 256 CUs at P >= 1e5.
// HBM-bound streaming kernels: forward reads `in` bytes per Gaussian and writes a 56-B projected
// record; backward re-reads the inputs, sums the Gaussian's gradient records and writes 248 B.
#include "sg_project.h"

template <int D>
__global__ void __launch_bounds__(256)
sg_preprocess_fwd_kernel(SgCam c, SgBatch bt, int P, const float *__restrict__ means3D,
                         const float *__restrict__ shs, const float *__restrict__ colors_precomp,
                         const float *__restrict__ opacities, const float *__restrict__ scales,
                         const float *__restrict__ rotations, const float *__restrict__ cov3D_precomp,
                         SgGeom g, SgBin bn, uint32_t cap, int32_t *__restrict__ radii, int hist_tiles, int nblocks)
{
    extern __shared__ uint32_t sg_hist_lds[];               // hist_tiles words (0: per-pair global atomics)
    __shared__ uint32_t scratch_all[4][256];
    // (Gaussian block, camera) of this workgroup: the K cameras of one block run back to back on ONE XCD (sg_block_frame)
    int gblock, frame;
    if (!sg_block_frame((int)blockIdx.x, bt.K, nblocks, gblock, frame)) return;
    c = sg_frame(c, frame, bt.cam_stride); g = sg_frame(g, (size_t)frame * bt.geom); bn = sg_frame(bn, (size_t)frame * bt.bin);
    radii += (size_t)frame * bt.P;
    const int idx = gblock * 256 + (int)threadIdx.x;
    const int wave = threadIdx.x >> 6;
    const bool live = idx < P;
    constexpr int nc = (D + 1) * (D + 1);
    float sh[nc * 3];
    if (shs && live) {
        const float *src = shs + (size_t)idx * c.M * 3;
        if (D == 3 && c.M == 16) {
            // the reference's layout [P,16,3]: a row is 192 B, 16-B aligned -> twelve 16-B loads per lane.  (Staging the
            // wave's 64 rows through LDS as coalesced 1-KiB pieces was measured too: 39.4 vs 39.0 us -- the one-row-per-lane
            // form does not bound this kernel; the pair expansion's scattered returning atomics do: 16 of its 39 us.)
#pragma unroll
            for (int k = 0; k < nc * 3 / 4; k++) {
                const float4 v = ((const float4 *)src)[k];
                sh[4 * k] = v.x; sh[4 * k + 1] = v.y; sh[4 * k + 2] = v.z; sh[4 * k + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < nc * 3; k++) sh[k] = src[k];
        }
    }
    SgProj o;
    o.mr = 0; o.tt = 0; o.clampbits = 0; o.x0 = o.y0 = o.x1 = o.y1 = 0;
    o.pix[0] = o.pix[1] = 0; o.conic[0] = o.conic[1] = o.conic[2] = 0; o.rgb[0] = o.rgb[1] = o.rgb[2] = 0; o.depth = 0;
    float opac = 0.0f;
    if (live) {
        float p[3] = { means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2] };
        float s3[3] = { 0, 0, 0 }, q[4] = { 0, 0, 0, 0 };
        if (!cov3D_precomp) {
            s3[0] = scales[3 * idx]; s3[1] = scales[3 * idx + 1]; s3[2] = scales[3 * idx + 2];
            q[0] = rotations[4 * idx]; q[1] = rotations[4 * idx + 1]; q[2] = rotations[4 * idx + 2]; q[3] = rotations[4 * idx + 3];
        }
        sg_project_fwd<D>(c, p, s3, q, cov3D_precomp ? cov3D_precomp + 6 * (size_t)idx : nullptr,
                          colors_precomp ? colors_precomp + 3 * (size_t)idx : nullptr, sh, o);
        opac = opacities[idx];
    }
    sg_store_proj(live, idx, o, opac, g, bn, c.gx, cap, radii, scratch_all[wave], hist_tiles ? sg_hist_lds : nullptr, hist_tiles,
                  sg_key_pitch_of(hist_tiles != 0, c.flags));
}

void sg_launch_preprocess_fwd(const SgCam &c, const SgBatch &bt, int P, const float *means3D, const float *shs,
                              const float *colors_precomp, const float *opacities, const float *scales,
                              const float *rotations, const float *cov3D_precomp, SgGeom g, SgBin b,
                              size_t cap, int32_t *radii, hipStream_t st)
{
    if (P <= 0) return;
    const int nblocks = (P + 255) / 256;
    dim3 grid(sg_frame_grid(nblocks, bt.K)), block(256);
    const int ht = sg_lds_hist(c.gx, c.gy) ? (int)sg_ctr_count((uint32_t)c.gx, (uint32_t)c.gy) : 0;     // histogram words (= tile counters)
#define SG_PP(DD) hipLaunchKernelGGL(sg_preprocess_fwd_kernel<DD>, grid, block, (size_t)ht * 4, st, c, bt, P, means3D, shs, \
                                     colors_precomp, opacities, scales, rotations, cov3D_precomp, g, b, sg_cap32(cap), radii, ht, nblocks)
    int D = colors_precomp ? 0 : c.D;
    sg_prof_begin(SG_K_PREPROCESS_FWD, st);
    switch (D) { case 0: SG_PP(0); break; case 1: SG_PP(1); break; case 2: SG_PP(2); break; default: SG_PP(3); break; }
    sg_prof_end(SG_K_PREPROCESS_FWD, st);
#undef SG_PP
}

// ------------------------------------------------------------------------------------------
// ACC: add to the gradient outputs instead of writing them (a compile-time switch: as a runtime flag the untaken branches cost
// the plain kernel 3.8 us of its 33 -- same-box A/B)
template <int D, bool ACC>
__global__ void __launch_bounds__(256)
sg_preprocess_bwd_kernel(SgCam c, int P, const float *__restrict__ means3D, const float *__restrict__ shs,
                         const float *__restrict__ colors_precomp, const float *__restrict__ scales,
                         const float *__restrict__ rotations, const float *__restrict__ cov3D_precomp,
                         const int32_t *__restrict__ radii, SgGeom g, SgRec grec,
                         size_t cap, const uint32_t *__restrict__ header, const uint8_t *__restrict__ rec_valid, float *__restrict__ dL_dmeans3D, float *__restrict__ dL_dmeans2D,
                         float *__restrict__ dL_dsh, float *__restrict__ dL_dcolors,
                         float *__restrict__ dL_dopacity, float *__restrict__ dL_dscales,
                         float *__restrict__ dL_drots, float *__restrict__ dL_dcov3D)
{
    constexpr bool accumulate = ACC;
    __shared__ float lds_all[4][32 * SG_ROW_LDS];         // 6.5 KiB per wave: record chunks, then dL/dsh rows out
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g0 = idx - lane;
    if (g0 >= P) return;                                    // whole wave out of range
    float *L = lds_all[wave];
    const bool live = idx < P;
    SgGaussGrad G;
#pragma unroll
    for (int k = 0; k < 3; k++) { G.dmean[k] = 0; G.dcol[k] = 0; G.dsc[k] = 0; }
#pragma unroll
    for (int k = 0; k < 4; k++) G.drot[k] = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) G.g6[k] = 0;
    G.g2[0] = G.g2[1] = 0; G.dop = 0;
    // after a forward that overflowed (header[1] != 0) the backward composite wrote no records: every gradient is ZERO, the
    // stale contents of the record buffer are never summed (asynchronous overflow check: rasterizer.py)
    const bool vis = live && radii[idx] > 0 && header[1] == 0u;
    const int Mrows = c.M;
    constexpr int nc = (D + 1) * (D + 1);
    const bool staged = D == 3 && Mrows == 16 && shs != nullptr && dL_dsh != nullptr;
    // 0. issue every load that does not depend on another one up front (one memory round trip, not four)
    float4 rc = make_float4(0, 0, 0, 0);
    float p[3] = { 0, 0, 0 }, s3[3] = { 0, 0, 0 }, q[4] = { 0, 0, 0, 0 }, sh[nc * 3], dsh[nc * 3];
    uint32_t flags = 0;
#pragma unroll
    for (int k = 0; k < nc * 3; k++) { sh[k] = 0.0f; dsh[k] = 0.0f; }
    if (vis) {
        { const uint2 sl = g.slot[idx]; rc.y = __uint_as_float(sl.x); rc.w = __uint_as_float(sl.y); }
        flags = g.flags[idx];
        p[0] = means3D[3 * idx]; p[1] = means3D[3 * idx + 1]; p[2] = means3D[3 * idx + 2];
        if (!cov3D_precomp) {
            s3[0] = scales[3 * idx]; s3[1] = scales[3 * idx + 1]; s3[2] = scales[3 * idx + 2];
            q[0] = rotations[4 * idx]; q[1] = rotations[4 * idx + 1]; q[2] = rotations[4 * idx + 2]; q[3] = rotations[4 * idx + 3];
        }
        if (shs) {
            const float *src = shs + (size_t)idx * Mrows * 3;
            if (staged) {
#pragma unroll
                for (int k = 0; k < nc * 3 / 4; k++) {
                    float4 v = ((const float4 *)src)[k];
                    sh[4 * k] = v.x; sh[4 * k + 1] = v.y; sh[4 * k + 2] = v.z; sh[4 * k + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int k = 0; k < nc * 3; k++) sh[k] = src[k];
            }
        }
    }
    // 1. this Gaussian's gradient records (wave-cooperative, coalesced)
    float a9[9];
    sg_sum_records_coop(grec, cap, vis, rc, lane, L, a9, rec_valid);
    // 2. the chain rule
    if (vis)
        sg_project_bwd<D>(c, p, s3, q, cov3D_precomp ? cov3D_precomp + 6 * (size_t)idx : nullptr, sh, flags, a9,
                          dL_dsh != nullptr, dsh, G);
    // 3. dL/dsh out: coefficient-major planes [M][P][3], only the (D+1)^2 in use (SG_FLAG_SH_PLANAR); or the reference's rows
    //    (every one of the M rows is written; coalesced through LDS when staged)
    if (dL_dsh && (c.flags & SG_FLAG_SH_PLANAR)) {
        if (live) {
#pragma unroll
            for (int kq = 0; kq < nc; kq++)
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    float *d = dL_dsh + ((size_t)kq * P + idx) * 3 + ch;
                    *d = accumulate ? dsh[3 * kq + ch] + *d : dsh[3 * kq + ch];
                }
        }
    } else if (dL_dsh) {
        if (staged) {
#pragma unroll
            for (int h = 0; h < 2; h++) {                    // 32 rows at a time (keeps LDS at 6.5 KiB per wave)
                if ((lane >> 5) == h) {
#pragma unroll
                    for (int k = 0; k < 12; k++)
                        *(float4 *)(L + (lane & 31) * SG_ROW_LDS + 4 * k) = make_float4(dsh[4 * k], dsh[4 * k + 1], dsh[4 * k + 2], dsh[4 * k + 3]);
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_wave_barrier();
                sg_rows48_store(dL_dsh, g0 + 32 * h, P, lane, L, 32, accumulate != 0);
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_wave_barrier();
            }
        } else if (live) {
            float *dsh_row = dL_dsh + (size_t)idx * Mrows * 3;
            if (accumulate) {
#pragma unroll
                for (int k = 0; k < nc * 3; k++) dsh_row[k] += dsh[k];
            } else {
#pragma unroll
                for (int k = 0; k < nc * 3; k++) dsh_row[k] = dsh[k];
                for (int k = nc * 3; k < Mrows * 3; k++) dsh_row[k] = 0.0f;
            }
        }
    }
    if (!live) return;
    // the screen-space gradient is per VIEW (the densifier's statistic): never accumulated
    dL_dmeans2D[3 * idx] = G.g2[0]; dL_dmeans2D[3 * idx + 1] = G.g2[1]; dL_dmeans2D[3 * idx + 2] = 0.0f;
    if (accumulate) {
        // the views of one optimisation step share ONE gradient buffer: this view adds to what the views in front of it in the
        // step's chain left there (the old values are requested together, here: one more memory round trip per wave)
        float o3[3], os[3], orr[4], oc[3], og[6], oo;
#pragma unroll
        for (int k = 0; k < 3; k++) o3[k] = dL_dmeans3D[3 * idx + k];
        oo = dL_dopacity[idx];
#pragma unroll
        for (int k = 0; k < 3; k++) { oc[k] = dL_dcolors ? dL_dcolors[3 * idx + k] : 0.0f; os[k] = dL_dscales ? dL_dscales[3 * idx + k] : 0.0f; }
#pragma unroll
        for (int k = 0; k < 4; k++) orr[k] = dL_drots ? dL_drots[4 * idx + k] : 0.0f;
#pragma unroll
        for (int k = 0; k < 6; k++) og[k] = dL_dcov3D ? dL_dcov3D[6 * idx + k] : 0.0f;
#pragma unroll
        for (int k = 0; k < 3; k++) { G.dmean[k] += o3[k]; G.dcol[k] += oc[k]; G.dsc[k] += os[k]; }
#pragma unroll
        for (int k = 0; k < 4; k++) G.drot[k] += orr[k];
#pragma unroll
        for (int k = 0; k < 6; k++) G.g6[k] += og[k];
        G.dop += oo;
    }
    dL_dmeans3D[3 * idx] = G.dmean[0]; dL_dmeans3D[3 * idx + 1] = G.dmean[1]; dL_dmeans3D[3 * idx + 2] = G.dmean[2];
    dL_dopacity[idx] = G.dop;
    if (dL_dcolors) { dL_dcolors[3 * idx] = G.dcol[0]; dL_dcolors[3 * idx + 1] = G.dcol[1]; dL_dcolors[3 * idx + 2] = G.dcol[2]; }
    if (dL_dscales) { dL_dscales[3 * idx] = G.dsc[0]; dL_dscales[3 * idx + 1] = G.dsc[1]; dL_dscales[3 * idx + 2] = G.dsc[2]; }
    if (dL_drots) { dL_drots[4 * idx] = G.drot[0]; dL_drots[4 * idx + 1] = G.drot[1]; dL_drots[4 * idx + 2] = G.drot[2]; dL_drots[4 * idx + 3] = G.drot[3]; }
    if (dL_dcov3D) {
#pragma unroll
        for (int k = 0; k < 6; k++) dL_dcov3D[6 * idx + k] = G.g6[k];
    }
}

// ---- K cameras per launch (round 4) -------------------------------------------------------------------------------------------
// K cameras (bt.K frames) of the SAME Gaussians: the kernel walks the frames -- chain rule with camera f on the record sums of frame
// f -- and sums the K gradients of a Gaussian in frame order (frame 0 assigns, frame f > 0 adds: bit for bit what K single-camera
// calls leave behind when the first writes the gradient buffer and the others run with accumulate = 1), then writes the 248-byte
// gradient row ONCE: K - 1 read-modify-write passes over the buffer less (at cfg3 2 x 47 MB per view), and the Gaussian's position,
// scale and rotation are read once for the K cameras.  dL_dmeans2D (the densifier's per-view statistic) is per frame.
//
// What a frame needs is kept OUT of the loop-carried registers (the first form of this kernel held 48 SH coefficients, 48 dL/dsh
// sums and the record-chunk staging across the loop: 316 registers at degree 3, one wave per SIMD, 45.8 us per view against the
// single-camera kernel's 30):
//  * the record sums of all frames come from sg_record_sums_kernel (a9 [K][P][12]: a kernel of ~40 registers at full occupancy
//    does the dependent chunk loop); the loop reads 48 contiguous bytes per Gaussian and frame, requested one frame ahead;
//  * the dL/dsh sums live in LDS ([coefficient][thread], pitch 257: conflict-free for the sums and for the transposed read of the
//    row store) -- 48 KiB per workgroup at degree 3: three workgroups per CU, which is what 168 registers allow anyway;
//  * the SH coefficients are re-read per frame (192 B per Gaussian out of L2: frame 0 brought them in).
#define SG_ACC_PITCH 257
template <int D, bool ACC>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, D >= 2 ? 2 : 3)))
sg_preprocess_bwd_frames_kernel(SgCam c0, SgBatch bt, int P, const float *__restrict__ means3D, const float *__restrict__ shs,
                         const float *__restrict__ colors_precomp, const float *__restrict__ scales,
                         const float *__restrict__ rotations, const float *__restrict__ cov3D_precomp,
                         const int32_t *__restrict__ radii0, SgGeom g0_, const uint32_t *__restrict__ header0,
                         const float4 *__restrict__ a9in, float *__restrict__ dL_dmeans3D, float *__restrict__ dL_dmeans2D,
                         float *__restrict__ dL_dsh, float *__restrict__ dL_dcolors,
                         float *__restrict__ dL_dopacity, float *__restrict__ dL_dscales,
                         float *__restrict__ dL_drots, float *__restrict__ dL_dcov3D)
{
    constexpr bool accumulate = ACC;
    extern __shared__ float sAcc[];                         // dL/dsh sums [nc * 3][SG_ACC_PITCH]; afterwards the staging rows of the store
    const int idx_all = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane_all = threadIdx.x & 63, wave_all = threadIdx.x >> 6;
    const int g0 = idx_all - lane_all;
    const bool live = idx_all < P;
    const int Mrows = c0.M;
    const int nframes = bt.K;
    constexpr int nc = (D + 1) * (D + 1);
    const bool want_sh = dL_dsh != nullptr;
    const bool planar = c0.flags & SG_FLAG_SH_PLANAR;
    const bool vec_rows = D == 3 && Mrows == 16 && shs != nullptr;            // 16-byte loads of the coefficient row
    const bool staged = vec_rows && want_sh && !planar;
    // the Gaussian's position / scale / rotation: read ONCE for all frames
    float p[3] = { 0, 0, 0 }, s3[3] = { 0, 0, 0 }, q[4] = { 0, 0, 0, 0 };
    // frame f + 1's (visible, clamp flags, record sums) are requested while frame f is worked on; frame 0's go out with the inputs
    bool vis_n = live && radii0[idx_all] > 0 && header0[1] == 0u;
    uint32_t fl_n = 0u;
    float4 n0 = make_float4(0, 0, 0, 0), n1 = n0, n2 = n0;
    if (vis_n) {
        fl_n = g0_.flags[idx_all];
        const float4 *src = a9in + 3 * (size_t)idx_all;
        n0 = src[0]; n1 = src[1]; n2 = src[2];
    }
    if (live) {
        p[0] = means3D[3 * idx_all]; p[1] = means3D[3 * idx_all + 1]; p[2] = means3D[3 * idx_all + 2];
        if (!cov3D_precomp) {
            s3[0] = scales[3 * idx_all]; s3[1] = scales[3 * idx_all + 1]; s3[2] = scales[3 * idx_all + 2];
            q[0] = rotations[4 * idx_all]; q[1] = rotations[4 * idx_all + 1]; q[2] = rotations[4 * idx_all + 2]; q[3] = rotations[4 * idx_all + 3];
        }
    }
    SgGaussGrad A;                                          // the sum over the frames
#pragma unroll
    for (int k = 0; k < 3; k++) { A.dmean[k] = 0; A.dcol[k] = 0; A.dsc[k] = 0; }
#pragma unroll
    for (int k = 0; k < 4; k++) A.drot[k] = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) A.g6[k] = 0;
    A.g2[0] = A.g2[1] = 0; A.dop = 0;
    // K > 1 and accumulate: the sums START from what the gradient buffer holds (read once, here), so that the K frames are added
    // in the order K single-camera calls with accumulate = 1 would add them: ((old + g0) + g1) + ...
    constexpr bool preload = ACC;
    if (preload && live) {
#pragma unroll
        for (int k = 0; k < 3; k++) A.dmean[k] = dL_dmeans3D[3 * idx_all + k];
        A.dop = dL_dopacity[idx_all];
#pragma unroll
        for (int k = 0; k < 3; k++) { A.dcol[k] = dL_dcolors ? dL_dcolors[3 * idx_all + k] : 0.0f; A.dsc[k] = dL_dscales ? dL_dscales[3 * idx_all + k] : 0.0f; }
#pragma unroll
        for (int k = 0; k < 4; k++) A.drot[k] = dL_drots ? dL_drots[4 * idx_all + k] : 0.0f;
#pragma unroll
        for (int k = 0; k < 6; k++) A.g6[k] = dL_dcov3D ? dL_dcov3D[6 * idx_all + k] : 0.0f;
    }
    if (want_sh) {
        float *acc = sAcc + threadIdx.x;
        if (preload && live) {
            const float *row = dL_dsh + (size_t)idx_all * Mrows * 3;
#pragma unroll
            for (int k = 0; k < nc * 3; k++) acc[k * SG_ACC_PITCH] = planar ? dL_dsh[((size_t)(k / 3) * P + idx_all) * 3 + k % 3] : row[k];
        } else {
#pragma unroll
            for (int k = 0; k < nc * 3; k++) acc[k * SG_ACC_PITCH] = 0.0f;
        }
    }
#pragma unroll 1
    for (int f = 0; f < nframes; f++) {
        // (opaque per trip: keeps hipcc from hoisting what depends on the thread's index only -- the SH row, LDS addresses, the
        //  per-frame arrays' addresses -- in front of the loop, where it would stay live across it: see sg_skin_bwd_frames_kernel)
        int idx = idx_all, tid = (int)threadIdx.x;
        asm volatile("" : "+v"(idx), "+v"(tid));
        const SgCam c = sg_frame(c0, f, bt.cam_stride);
        SgGaussGrad G;
#pragma unroll
        for (int k = 0; k < 3; k++) { G.dmean[k] = 0; G.dcol[k] = 0; G.dsc[k] = 0; }
#pragma unroll
        for (int k = 0; k < 4; k++) G.drot[k] = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) G.g6[k] = 0;
        G.g2[0] = G.g2[1] = 0; G.dop = 0;
        // after a forward that overflowed (header[1] != 0) the backward composite wrote no records: every gradient is ZERO, the
        // stale contents of the record buffer are never summed (asynchronous overflow check: rasterizer.py)
        const bool vis = vis_n;
        const uint32_t flags = fl_n;
        const float a9[9] = { n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w, n2.x };
        if (f + 1 < nframes) {
            const SgGeom gn = sg_frame(g0_, (size_t)(f + 1) * bt.geom);
            vis_n = live && radii0[(size_t)(f + 1) * bt.P + idx] > 0 && sg_at(header0, (size_t)(f + 1) * bt.bin)[1] == 0u;
            fl_n = 0u;
            if (vis_n) {
                fl_n = gn.flags[idx];
                const float4 *src = a9in + 3 * ((size_t)(f + 1) * bt.P + idx);
                n0 = src[0]; n1 = src[1]; n2 = src[2];
            }
        }
        // the chain rule; dL/dsh rows: frame 0 assigns, later frames add
        if (vis) {
            float sh[nc * 3], dshf[nc * 3];
#pragma unroll
            for (int k = 0; k < nc * 3; k++) { sh[k] = 0.0f; dshf[k] = 0.0f; }
            if (shs) {
                const float *src = shs + (size_t)idx * Mrows * 3;
                if (vec_rows) {
#pragma unroll
                    for (int k = 0; k < nc * 3 / 4; k++) {
                        float4 v = ((const float4 *)src)[k];
                        sh[4 * k] = v.x; sh[4 * k + 1] = v.y; sh[4 * k + 2] = v.z; sh[4 * k + 3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < nc * 3; k++) sh[k] = src[k];
                }
            }
            sg_project_bwd<D>(c, p, s3, q, cov3D_precomp ? cov3D_precomp + 6 * (size_t)idx : nullptr, sh, flags, a9,
                              want_sh, dshf, G);
            if (want_sh) {
                float *acc = sAcc + tid;
                if (!preload && f == 0) {
#pragma unroll
                    for (int k = 0; k < nc * 3; k++) acc[k * SG_ACC_PITCH] = dshf[k];
                } else {
#pragma unroll
                    for (int k = 0; k < nc * 3; k++) acc[k * SG_ACC_PITCH] = acc[k * SG_ACC_PITCH] + dshf[k];
                }
            }
        }
        // the screen-space gradient is per VIEW (the densifier's statistic): never accumulated
        if (live) {
            float *m2 = dL_dmeans2D + 3 * ((size_t)f * bt.P + idx);
            m2[0] = G.g2[0]; m2[1] = G.g2[1]; m2[2] = 0.0f;
        }
        if (!preload && f == 0) {
            A = G;
        } else {
            // (operand order of the single-camera accumulate: new + old)
#pragma unroll
            for (int k = 0; k < 3; k++) { A.dmean[k] = G.dmean[k] + A.dmean[k]; A.dcol[k] = G.dcol[k] + A.dcol[k]; A.dsc[k] = G.dsc[k] + A.dsc[k]; }
#pragma unroll
            for (int k = 0; k < 4; k++) A.drot[k] = G.drot[k] + A.drot[k];
#pragma unroll
            for (int k = 0; k < 6; k++) A.g6[k] = G.g6[k] + A.g6[k];
            A.dop = G.dop + A.dop;
        }
    }
    SgGaussGrad &G = A;
    const int idx = idx_all, lane = lane_all;
    // dL/dsh out: coefficient-major planes [M][P][3], only the (D+1)^2 in use (SG_FLAG_SH_PLANAR); or the reference's rows
    // (every one of the M rows is written; coalesced through LDS when staged).  (accumulate: the sums already hold old + frames)
    if (want_sh && planar) {
        if (live) {
            const float *acc = sAcc + threadIdx.x;
#pragma unroll
            for (int k = 0; k < nc * 3; k++) dL_dsh[((size_t)(k / 3) * P + idx) * 3 + k % 3] = acc[k * SG_ACC_PITCH];
        }
    } else if (want_sh) {
        float dsh[nc * 3];
        {
            const float *acc = sAcc + threadIdx.x;
#pragma unroll
            for (int k = 0; k < nc * 3; k++) dsh[k] = acc[k * SG_ACC_PITCH];
        }
        if (staged) {
            __syncthreads();                                 // every thread holds its sums: the LDS is free for the staging rows
            float *L = sAcc + wave_all * (32 * SG_ROW_LDS);
            static_assert(4 * 32 * SG_ROW_LDS <= 48 * SG_ACC_PITCH, "staging rows fit the accumulator block");
#pragma unroll
            for (int h = 0; h < 2; h++) {                    // 32 rows at a time
                if ((lane >> 5) == h) {
#pragma unroll
                    for (int k = 0; k < 12; k++)
                        *(float4 *)(L + (lane & 31) * SG_ROW_LDS + 4 * k) = make_float4(dsh[4 * k], dsh[4 * k + 1], dsh[4 * k + 2], dsh[4 * k + 3]);
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_wave_barrier();
                if (g0 < P) sg_rows48_store(dL_dsh, g0 + 32 * h, P, lane, L, 32, false);
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_wave_barrier();
            }
        } else if (live) {
            float *dsh_row = dL_dsh + (size_t)idx * Mrows * 3;
#pragma unroll
            for (int k = 0; k < nc * 3; k++) dsh_row[k] = dsh[k];
            if (!accumulate)
                for (int k = nc * 3; k < Mrows * 3; k++) dsh_row[k] = 0.0f;
        }
    }
    if (!live) return;
    dL_dmeans3D[3 * idx] = G.dmean[0]; dL_dmeans3D[3 * idx + 1] = G.dmean[1]; dL_dmeans3D[3 * idx + 2] = G.dmean[2];
    dL_dopacity[idx] = G.dop;
    if (dL_dcolors) { dL_dcolors[3 * idx] = G.dcol[0]; dL_dcolors[3 * idx + 1] = G.dcol[1]; dL_dcolors[3 * idx + 2] = G.dcol[2]; }
    if (dL_dscales) { dL_dscales[3 * idx] = G.dsc[0]; dL_dscales[3 * idx + 1] = G.dsc[1]; dL_dscales[3 * idx + 2] = G.dsc[2]; }
    if (dL_drots) { dL_drots[4 * idx] = G.drot[0]; dL_drots[4 * idx + 1] = G.drot[1]; dL_drots[4 * idx + 2] = G.drot[2]; dL_drots[4 * idx + 3] = G.drot[3]; }
    if (dL_dcov3D) {
#pragma unroll
        for (int k = 0; k < 6; k++) dL_dcov3D[6 * idx + k] = G.g6[k];
    }
}

void sg_launch_preprocess_bwd(const SgCam &c, const SgBatch &bt, int P, const float *means3D, const float *shs,
                              const float *colors_precomp, const float *opacities, const float *scales,
                              const float *rotations, const float *cov3D_precomp,
                              const int32_t *radii, SgGeom g, SgRec grec, size_t cap, const uint32_t *header, const uint8_t *rec_valid,
                              float4 *a9buf, float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dsh,
                              float *dL_dcolors, float *dL_dopacity, float *dL_dscales,
                              float *dL_drots, float *dL_dcov3D, int accumulate, hipStream_t st)
{
    (void)opacities;
    if (P <= 0) return;
    dim3 grid((P + 255) / 256), block(256);
    int D = shs ? c.D : 0;
    // K > 1: dL/dsh sums in LDS ([coefficient][thread]); the staged row store reuses the block
    const size_t dyn = dL_dsh ? (size_t)(D + 1) * (D + 1) * 3 * SG_ACC_PITCH * sizeof(float) : 0;
#define SG_PB1(DD, AA) hipLaunchKernelGGL((sg_preprocess_bwd_kernel<DD, AA>), grid, block, 0, st, c, P, means3D, shs, \
                                     colors_precomp, scales, rotations, cov3D_precomp, radii, g,            \
                                     grec, cap, header, rec_valid, dL_dmeans3D, dL_dmeans2D, dL_dsh, dL_dcolors, \
                                     dL_dopacity, dL_dscales, dL_drots, dL_dcov3D)
#define SG_PBK(DD, AA) hipLaunchKernelGGL((sg_preprocess_bwd_frames_kernel<DD, AA>), grid, block, dyn, st, c, bt, P, means3D, shs, \
                                     colors_precomp, scales, rotations, cov3D_precomp, radii, g,            \
                                     header, a9buf, dL_dmeans3D, dL_dmeans2D, dL_dsh, dL_dcolors, \
                                     dL_dopacity, dL_dscales, dL_drots, dL_dcov3D)
#define SG_PB2(DD, AA) do { if (bt.K == 1) SG_PB1(DD, AA); else SG_PBK(DD, AA); } while (0)      // (K = 1: the round-3 kernel, untouched)
#define SG_PB(DD) do { if (accumulate) SG_PB2(DD, true); else SG_PB2(DD, false); } while (0)
    sg_prof_begin(SG_K_PREPROCESS_BWD, st);
    // K > 1: the record sums of all frames first, as a kernel of their own (a9 [K][P][12] behind the K record buffers)
    if (bt.K > 1) sg_launch_record_sums(bt, P, radii, g, grec, cap, header, rec_valid, a9buf, st);
    switch (D) { case 0: SG_PB(0); break; case 1: SG_PB(1); break; case 2: SG_PB(2); break; default: SG_PB(3); break; }
    sg_prof_end(SG_K_PREPROCESS_BWD, st);
#undef SG_PB
#undef SG_PB2
#undef SG_PB1
#undef SG_PBK
}
