// SMPL linear-blend-skinning fused with the projection preprocess (forward and backward).
//
// Replaces the LBS block of SinGS.forward -- sings/rec/models/sings_hybrid.py:398-428 with
// sings/rec/utils/body_model/lbs.py:59-74 (T = W . A, v' = T [v;1]) and
// sings/rec/utils/geometry/rotations.py:98-149 (matrix_to_quaternion) -- and feeds the posed
// mean / rotation / scale straight into the projection of sg_project.h, so the posed means and
// quaternions, T[N,4,4] and the ~15 eager kernels of the reference never touch HBM.
//
// The only dense contraction on the path, T[64 x 16] = W[64 x J] . A[J x 16] per wave, runs on the
// matrix cores with v_mfma_f32_16x16x4_f32 (exact fp32, bitwise a k-ordered fmaf chain); its
// transpose dA[J x 16] = W^T . dT in backward likewise.  W is streamed once per pass in 64-B row
// segments through LDS (HBM-bound: 4J bytes per Gaussian dominate).
#include "sg_project.h"
#include "sg_rot.h"

#define SG_SKIN_THREADS 256
#define SG_SKIN_WAVES 4
#define SG_JMAX 64
#define SG_WSTRIDE 17

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SgSkin {
    int J, rot6d;            // rot6d: rot_canon is [P,6] (Zhou et al. 6-D form, rotation_6d_to_matrix fused in)
    const float *xyz_canon, *rot_canon, *lbs_w, *A, *smpl_scale, *transl, *ext_trans, *ext_rot, *ext_scale;
};

// ---- rotations.py:98-149 ----------------------------------------------------------------
__device__ __forceinline__ void sg_m2q(const float m[9], float q[4], int &best, float &qab)
{
    float x0 = 1.0f + m[0] + m[4] + m[8], x1 = 1.0f + m[0] - m[4] - m[8];
    float x2 = 1.0f - m[0] + m[4] - m[8], x3 = 1.0f - m[0] - m[4] + m[8];
    float a0 = x0 > 0.0f ? sqrtf(x0) : 0.0f, a1 = x1 > 0.0f ? sqrtf(x1) : 0.0f;
    float a2 = x2 > 0.0f ? sqrtf(x2) : 0.0f, a3 = x3 > 0.0f ? sqrtf(x3) : 0.0f;
    best = 0; qab = a0;
    if (a1 > qab) { best = 1; qab = a1; }
    if (a2 > qab) { best = 2; qab = a2; }
    if (a3 > qab) { best = 3; qab = a3; }
    float n0, n1, n2, n3;
    if (best == 0)      { n0 = a0 * a0;      n1 = m[7] - m[5]; n2 = m[2] - m[6]; n3 = m[3] - m[1]; }
    else if (best == 1) { n0 = m[7] - m[5];  n1 = a1 * a1;     n2 = m[3] + m[1]; n3 = m[2] + m[6]; }
    else if (best == 2) { n0 = m[2] - m[6];  n1 = m[3] + m[1]; n2 = a2 * a2;     n3 = m[5] + m[7]; }
    else                { n0 = m[3] - m[1];  n1 = m[6] + m[2]; n2 = m[7] + m[5]; n3 = a3 * a3; }
    float den = 2.0f * (qab > 0.1f ? qab : 0.1f);
    q[0] = n0 / den; q[1] = n1 / den; q[2] = n2 / den; q[3] = n3 / den;
}

// gradient of sg_m2q: dq[4] -> dm[9] (autograd of the reference expression: zero sub-gradient of
// sqrt at <= 0, of the 0.1 floor below it, and only the selected candidate receives gradient)
__device__ __forceinline__ void sg_m2q_bwd(const float q[4], int best, float qab, const float dq[4], float dm[9])
{
#pragma unroll
    for (int i = 0; i < 9; i++) dm[i] = 0.0f;
    float den = 2.0f * (qab > 0.1f ? qab : 0.1f);
    float dn0 = dq[0] / den, dn1 = dq[1] / den, dn2 = dq[2] / den, dn3 = dq[3] / den;
    float dden = -(dq[0] * q[0] + dq[1] * q[1] + dq[2] * q[2] + dq[3] * q[3]) / den;
    // d(num_best)/dx = 1 (x>0), d(den)/dx = 2 * [qab > 0.1] * 0.5 / qab
    float dx = 0.0f;
    if (qab > 0.0f) {
        float dnb = best == 0 ? dn0 : (best == 1 ? dn1 : (best == 2 ? dn2 : dn3));
        dx = dnb + (qab > 0.1f ? dden / qab : 0.0f);
    }
    if (best == 0) {
        dm[0] += dx; dm[4] += dx; dm[8] += dx;
        dm[7] += dn1; dm[5] -= dn1; dm[2] += dn2; dm[6] -= dn2; dm[3] += dn3; dm[1] -= dn3;
    } else if (best == 1) {
        dm[0] += dx; dm[4] -= dx; dm[8] -= dx;
        dm[7] += dn0; dm[5] -= dn0; dm[3] += dn2; dm[1] += dn2; dm[2] += dn3; dm[6] += dn3;
    } else if (best == 2) {
        dm[0] -= dx; dm[4] += dx; dm[8] -= dx;
        dm[2] += dn0; dm[6] -= dn0; dm[3] += dn1; dm[1] += dn1; dm[5] += dn3; dm[7] += dn3;
    } else {
        dm[0] -= dx; dm[4] -= dx; dm[8] += dx;
        dm[3] += dn0; dm[1] -= dn0; dm[6] += dn1; dm[2] += dn1; dm[7] += dn2; dm[5] += dn2;
    }
}

// rotations.py:372-407: Hamilton product, real part made non-negative
__device__ __forceinline__ void sg_qmul_std(const float a[4], const float b[4], float o[4])
{
    o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    if (o[0] < 0.0f) { o[0] = -o[0]; o[1] = -o[1]; o[2] = -o[2]; o[3] = -o[3]; }
}

// ---- W . A on the matrix cores -----------------------------------------------------------
// Stages joints [16c, 16c+16) of the wave's 64 skinning-weight rows into sW[64][17] (zero padded).
__device__ __forceinline__ void sg_stage_w_chunk(const float *__restrict__ W, int J, int P, int g0, int c, int lane,
                                                 float *__restrict__ sW)
{
    if ((J & 3) == 0) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int row = i * 16 + (lane >> 2), col = c * 16 + 4 * (lane & 3);
            float4 v = make_float4(0, 0, 0, 0);
            if (g0 + row < P && col < J) v = *(const float4 *)(W + (size_t)(g0 + row) * J + col);
            float *d = sW + row * SG_WSTRIDE + 4 * (lane & 3);
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            int row = i * 4 + (lane >> 4), col = c * 16 + (lane & 15);
            float v = 0.0f;
            if (g0 + row < P && col < J) v = W[(size_t)(g0 + row) * J + col];
            sW[row * SG_WSTRIDE + (lane & 15)] = v;
        }
    }
}

// the same in two halves (J % 4 == 0): issue the loads of a chunk / put them into LDS -- the chunk loops below keep the
// next chunk's loads in flight while the matrix cores work on the current one
__device__ __forceinline__ void sg_w_fetch(const float *__restrict__ W, int J, int P, int g0, int c, int lane, float4 v[4])
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = i * 16 + (lane >> 2), col = c * 16 + 4 * (lane & 3);
        v[i] = make_float4(0, 0, 0, 0);
        if (g0 + row < P && col < J) v[i] = *(const float4 *)(W + (size_t)(g0 + row) * J + col);
    }
}
__device__ __forceinline__ void sg_w_stash(const float4 v[4], int lane, float *__restrict__ sW)
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float *d = sW + (i * 16 + (lane >> 2)) * SG_WSTRIDE + 4 * (lane & 3);
        d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
    }
}

// T rows 0..2 (12 floats, row-major 3x4) of this lane's Gaussian g0 + lane.
// sA: [Jp][16] joint transforms (zero padded to a multiple of 16 rows), sW / sT: this wave's scratch.
__device__ __forceinline__ void sg_skin_T(const float *__restrict__ W, int J, int P, int g0, int lane,
                                          const float *__restrict__ sA, float *__restrict__ sW,
                                          float *__restrict__ sT, float T[12])
{
    f32x4 acc[4];
#pragma unroll
    for (int b = 0; b < 4; b++) acc[b] = (f32x4){ 0.0f, 0.0f, 0.0f, 0.0f };
    const int nchunk = (J + 15) >> 4;
    const bool vec = (J & 3) == 0;
    float4 wv[4];
    if (vec) sg_w_fetch(W, J, P, g0, 0, lane, wv);
    for (int c = 0; c < nchunk; c++) {
        if (vec) {
            sg_w_stash(wv, lane, sW);
            if (c + 1 < nchunk) sg_w_fetch(W, J, P, g0, c + 1, lane, wv);
            __builtin_amdgcn_s_waitcnt(0xC07F);             // LDS only: the next chunk's global loads stay in flight
        } else {
            sg_stage_w_chunk(W, J, P, g0, c, lane, sW);
            __builtin_amdgcn_s_waitcnt(0);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            float bv = sA[(c * 16 + 4 * kk + (lane >> 4)) * 16 + (lane & 15)];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                float av = sW[(16 * b + (lane & 15)) * SG_WSTRIDE + 4 * kk + (lane >> 4)];
                acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[b], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // acc[b][r] = T[gaussian 16b + 4(lane>>4) + r][entry lane&15]  ->  transpose through LDS
    if ((lane & 15) < 12) {
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) sT[(16 * b + 4 * (lane >> 4) + r) * 13 + (lane & 15)] = acc[b][r];
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 12; i++) T[i] = sT[lane * 13 + i];
    __builtin_amdgcn_wave_barrier();
}

struct SgPosed { float p[3], q[4], s3[3], Rdef[9], x[3], Rc[9], sc; int best; float qab; };

// sings_hybrid.py:400-428 for one Gaussian, given its blended transform T (3x4)
__device__ __forceinline__ void sg_pose_gaussian(const SgSkin &k, int idx, const float T[12],
                                                 const float *__restrict__ scales, SgPosed &o)
{
    o.x[0] = k.xyz_canon[3 * idx]; o.x[1] = k.xyz_canon[3 * idx + 1]; o.x[2] = k.xyz_canon[3 * idx + 2];
#pragma unroll
    for (int i = 0; i < 3; i++) o.p[i] = T[4 * i] * o.x[0] + T[4 * i + 1] * o.x[1] + T[4 * i + 2] * o.x[2] + T[4 * i + 3];
    o.s3[0] = scales[3 * idx]; o.s3[1] = scales[3 * idx + 1]; o.s3[2] = scales[3 * idx + 2];
    o.sc = k.smpl_scale ? k.smpl_scale[0] : 1.0f;
    if (k.smpl_scale) {
#pragma unroll
        for (int i = 0; i < 3; i++) { o.p[i] = o.p[i] * o.sc; o.s3[i] = o.s3[i] * o.sc; }
    }
    if (k.transl) {
#pragma unroll
        for (int i = 0; i < 3; i++) o.p[i] = o.p[i] + k.transl[i];
    }
    if (k.rot_canon) {
        if (k.rot6d) {       // sings_hybrid.py:356-357 fused in: the [N,3,3] matrices never exist in HBM
            float d6[6];
#pragma unroll
            for (int i = 0; i < 6; i++) d6[i] = k.rot_canon[6 * (size_t)idx + i];
            sg_r6d2m(d6, o.Rc);
        } else {
#pragma unroll
            for (int i = 0; i < 9; i++) o.Rc[i] = k.rot_canon[9 * (size_t)idx + i];
        }
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++)
                o.Rdef[3 * i + j] = T[4 * i] * o.Rc[j] + T[4 * i + 1] * o.Rc[3 + j] + T[4 * i + 2] * o.Rc[6 + j];
    } else {                    // isotropic: R_canon = I (sings_hybrid.py:358-361)
#pragma unroll
        for (int i = 0; i < 9; i++) o.Rc[i] = (i % 4 == 0) ? 1.0f : 0.0f;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) o.Rdef[3 * i + j] = T[4 * i + j];
    }
    sg_m2q(o.Rdef, o.q, o.best, o.qab);
    if (k.ext_rot) {            // sings_hybrid.py:421-428 (animation only; forward)
        float e = k.ext_scale[0], pr[3];
#pragma unroll
        for (int i = 0; i < 3; i++)
            pr[i] = k.ext_rot[3 * i] * o.p[0] + k.ext_rot[3 * i + 1] * o.p[1] + k.ext_rot[3 * i + 2] * o.p[2];
#pragma unroll
        for (int i = 0; i < 3; i++) { o.p[i] = k.ext_trans[i] + e * pr[i]; o.s3[i] = e * o.s3[i]; }
        float er[9], qe[4], qo[4]; int bb; float qq;
#pragma unroll
        for (int i = 0; i < 9; i++) er[i] = k.ext_rot[i];
        sg_m2q(er, qe, bb, qq);
        sg_qmul_std(qe, o.q, qo);
        o.q[0] = qo[0]; o.q[1] = qo[1]; o.q[2] = qo[2]; o.q[3] = qo[3];
    }
}

__device__ __forceinline__ void sg_load_A(const SgSkin &k, float *__restrict__ sA)
{
    const int Jp = ((k.J + 15) >> 4) << 4;
    for (int i = threadIdx.x; i < Jp * 16; i += blockDim.x) sA[i] = i < k.J * 16 ? k.A[i] : 0.0f;
}

// frame f of a launch of bt.K frames: joint transforms A + 16 J f, translation + transl_stride f (sg_frame for the rest)
__device__ __forceinline__ SgSkin sg_skin_frame(SgSkin k, int frame, const SgBatch &bt)
{
    k.A += (size_t)frame * k.J * 16;
    if (k.transl) k.transl += (size_t)frame * bt.transl_stride;
    return k;
}

template <int D>
__global__ void __launch_bounds__(SG_SKIN_THREADS)
sg_skin_fwd_kernel(SgCam c, SgBatch bt, int P, SgSkin k, const float *__restrict__ shs, const float *__restrict__ opacities,
                   const float *__restrict__ scales, SgGeom g, SgBin bn, uint32_t cap,
                   int32_t *__restrict__ radii, float *__restrict__ posed_xyz, float *__restrict__ posed_rotq,
                   float *__restrict__ posed_scales, int hist_tiles, int nblocks)
{
    __shared__ float sA[SG_JMAX * 16];
    __shared__ float sW[SG_SKIN_WAVES][64 * SG_WSTRIDE];
    __shared__ float sT[SG_SKIN_WAVES][64 * 13];
    static_assert(SG_SKIN_WAVES * 64 * SG_WSTRIDE >= SG_HIST_TILES_MAX, "the weight tiles double as the per-tile histogram");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // (Gaussian block, frame): the K poses of one block of 256 Gaussians run back to back on one XCD -- its skinning weights
    // (4 J bytes per Gaussian: the largest input), means, scales and SH rows come from HBM once and from that L2 K - 1 times
    int gblock, frame;
    if (!sg_block_frame((int)blockIdx.x, bt.K, nblocks, gblock, frame)) return;
    c = sg_frame(c, frame, bt.cam_stride); k = sg_skin_frame(k, frame, bt);
    g = sg_frame(g, (size_t)frame * bt.geom); bn = sg_frame(bn, (size_t)frame * bt.bin);
    radii += (size_t)frame * bt.P;
    if (posed_xyz) posed_xyz += 3 * (size_t)frame * bt.P;
    if (posed_rotq) posed_rotq += 4 * (size_t)frame * bt.P;
    if (posed_scales) posed_scales += 3 * (size_t)frame * bt.P;
    const int g0 = (gblock * SG_SKIN_WAVES + wave) * 64;
    const int idx = g0 + lane;
    sg_load_A(k, sA);
    __syncthreads();
    float T[12];
    sg_skin_T(k.lbs_w, k.J, P, g0, lane, sA, sW[wave], sT[wave], T);
    const bool live = idx < P;
    SgProj o;
    o.mr = 0; o.tt = 0; o.clampbits = 0; o.x0 = o.y0 = o.x1 = o.y1 = 0;
    o.pix[0] = o.pix[1] = 0; o.conic[0] = o.conic[1] = o.conic[2] = 0; o.rgb[0] = o.rgb[1] = o.rgb[2] = 0; o.depth = 0;
    float opac = 0.0f;
    if (live) {
        SgPosed ps;
        sg_pose_gaussian(k, idx, T, scales, ps);
        sg_project_fwd<D>(c, ps.p, ps.s3, ps.q, nullptr, nullptr, shs + (size_t)idx * c.M * 3, o);
        opac = opacities[idx];
        if (posed_xyz) { posed_xyz[3 * idx] = ps.p[0]; posed_xyz[3 * idx + 1] = ps.p[1]; posed_xyz[3 * idx + 2] = ps.p[2]; }
        if (posed_rotq) { posed_rotq[4 * idx] = ps.q[0]; posed_rotq[4 * idx + 1] = ps.q[1]; posed_rotq[4 * idx + 2] = ps.q[2]; posed_rotq[4 * idx + 3] = ps.q[3]; }
        if (posed_scales) { posed_scales[3 * idx] = ps.s3[0]; posed_scales[3 * idx + 1] = ps.s3[1]; posed_scales[3 * idx + 2] = ps.s3[2]; }
    }
    uint32_t *hist = nullptr;
    if (hist_tiles) {                                         // the skinning-weight tiles are dead: reuse them as histogram
        __syncthreads();
        hist = (uint32_t *)&sW[0][0];
    }
    sg_store_proj(live, idx, o, opac, g, bn, c.gx, cap, radii, (uint32_t *)sT[wave], hist, hist_tiles, sg_key_pitch_of(hist_tiles != 0, c.flags));
}

// Backward: LBS^T.  Per Gaussian: dL/dxyz_canon, dL/dR_canon, dL/dscales, dL/dopacity, dL/dsh;
// per workgroup: partial dL/dA [Jp x 16] (matrix cores) and dL/dtransl [3] written to a slab
// (reduced by sg_skin_reduce_kernel -- no atomics, deterministic).
template <int D, bool ACC>
// D = 0 (what SinGS trains at): three waves per SIMD.  Unconstrained hipcc takes 196 registers = two waves per SIMD = 2 048 resident
// waves, and an avatar of 150 k Gaussians is 2 344 waves: a second, 14 %-full round (the kernel is latency-bound: a round costs what
// a wave costs).  With the hint it fits 168 registers without scratch.  D >= 1 would spill (36-52 B at D = 1): left alone.
__global__ void __launch_bounds__(SG_SKIN_THREADS) __attribute__((amdgpu_waves_per_eu(D == 0 ? 3 : 1, D == 0 ? 3 : 10)))
sg_skin_bwd_kernel(SgCam c, int P, SgSkin k, const float *__restrict__ shs, const float *__restrict__ scales,
                   const int32_t *__restrict__ radii, SgGeom g, SgRec grec, size_t cap,
                   const uint32_t *__restrict__ header, const uint8_t *__restrict__ rec_valid, const float *__restrict__ dposed_xyz_in, const float *__restrict__ dposed_rotq_in,
                   float *__restrict__ dL_dxyz_canon, float *__restrict__ dL_drot_canon,
                   float *__restrict__ dL_dscales, float *__restrict__ dL_dopacity, float *__restrict__ dL_dsh,
                   float *__restrict__ dL_dmeans2D, float *__restrict__ slab, int slab_stride)
{
    constexpr bool accumulate = ACC;                            // (compile-time: see sg_preprocess_bwd_kernel)
    __shared__ float sA[SG_JMAX * 16];
    // per wave: [weights tile | T transpose scratch, later the dT tile]; between the two uses the whole 2 x 4.3 KB is
    // the staging buffer of the record sums and of the dL/dsh rows
    __shared__ float sWT[SG_SKIN_WAVES][2 * 64 * SG_WSTRIDE];
    static_assert(2 * 64 * SG_WSTRIDE >= SG_REC_CHUNK * 12 && 2 * 64 * SG_WSTRIDE >= 32 * SG_ROW_LDS, "staging buffer size");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *const sWw = sWT[wave], *const sTw = sWT[wave] + 64 * SG_WSTRIDE;
    const int g0 = (blockIdx.x * SG_SKIN_WAVES + wave) * 64;
    const int idx = g0 + lane;
    sg_load_A(k, sA);
    __syncthreads();
    float T[12];
    sg_skin_T(k.lbs_w, k.J, P, g0, lane, sA, sWw, sTw, T);
    const bool live = idx < P;
    float dT[12];
#pragma unroll
    for (int i = 0; i < 12; i++) dT[i] = 0.0f;
    float dtr[3] = { 0, 0, 0 };
    const int Mrows = c.M;
    constexpr int nc = (D + 1) * (D + 1);
    // this Gaussian's gradient records: wave-cooperative, coalesced (as in sg_preprocess_bwd_kernel)
    const bool vis = live && radii[idx] > 0 && header[1] == 0u;     // forward overflowed: zero gradients (see sg_preprocess.hip)
    float4 rc = make_float4(0, 0, 0, 0);
    if (vis) { const uint2 sl = g.slot[idx]; rc.y = __uint_as_float(sl.x); rc.w = __uint_as_float(sl.y); }
    float a9[9];
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    sg_sum_records_coop(grec, cap, vis, rc, lane, sWw, a9, rec_valid);
    float dsh[nc * 3];
#pragma unroll
    for (int i = 0; i < nc * 3; i++) dsh[i] = 0.0f;
    if (live) {
        float dxc[3] = { 0, 0, 0 }, dRc[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, dsc[3] = { 0, 0, 0 }, dop = 0, g2[2] = { 0, 0 };
        const bool have_in = dposed_xyz_in != nullptr || dposed_rotq_in != nullptr;
        if (vis || have_in) {
            SgPosed ps;
            sg_pose_gaussian(k, idx, T, scales, ps);
            SgGaussGrad G;
#pragma unroll
            for (int i = 0; i < 3; i++) { G.dmean[i] = 0; G.dsc[i] = 0; G.dcol[i] = 0; }
#pragma unroll
            for (int i = 0; i < 4; i++) G.drot[i] = 0;
            G.g2[0] = G.g2[1] = 0; G.dop = 0;
            if (vis)
                sg_project_bwd<D>(c, ps.p, ps.s3, ps.q, nullptr, shs + (size_t)idx * Mrows * 3, g.flags[idx], a9, true, dsh, G);
            if (dposed_xyz_in) {
#pragma unroll
                for (int i = 0; i < 3; i++) G.dmean[i] += dposed_xyz_in[3 * idx + i];
            }
            if (dposed_rotq_in) {
#pragma unroll
                for (int i = 0; i < 4; i++) G.drot[i] += dposed_rotq_in[4 * idx + i];
            }
            g2[0] = G.g2[0]; g2[1] = G.g2[1]; dop = G.dop;
            // scales_posed = scales * s ; p = (T33 x + t) * s + transl
#pragma unroll
            for (int i = 0; i < 3; i++) { dsc[i] = G.dsc[i] * ps.sc; dtr[i] = G.dmean[i]; }
            float dps[3] = { G.dmean[0] * ps.sc, G.dmean[1] * ps.sc, G.dmean[2] * ps.sc };
            float dRd[9];
            sg_m2q_bwd(ps.q, ps.best, ps.qab, G.drot, dRd);
#pragma unroll
            for (int i = 0; i < 3; i++) {
#pragma unroll
                for (int kx = 0; kx < 3; kx++) {
                    // dT33 = dp x^T + dRdef Rc^T
                    dT[4 * i + kx] = dps[i] * ps.x[kx] + dRd[3 * i] * ps.Rc[3 * kx] + dRd[3 * i + 1] * ps.Rc[3 * kx + 1] + dRd[3 * i + 2] * ps.Rc[3 * kx + 2];
                }
                dT[4 * i + 3] = dps[i];
            }
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
                dxc[kx] = T[kx] * dps[0] + T[4 + kx] * dps[1] + T[8 + kx] * dps[2];
#pragma unroll
                for (int j = 0; j < 3; j++)      // dRc = T33^T dRdef
                    dRc[3 * kx + j] = T[kx] * dRd[j] + T[4 + kx] * dRd[3 + j] + T[8 + kx] * dRd[6 + j];
            }
        }
        // accumulate: the frames of one optimisation step share ONE canonical-gradient buffer (sg_skinned_backward_gaussians)
        if (accumulate) {
#pragma unroll
            for (int i = 0; i < 3; i++) { dxc[i] += dL_dxyz_canon[3 * idx + i]; dsc[i] += dL_dscales[3 * idx + i]; }
            dop += dL_dopacity[idx];
        }
        dL_dxyz_canon[3 * idx] = dxc[0]; dL_dxyz_canon[3 * idx + 1] = dxc[1]; dL_dxyz_canon[3 * idx + 2] = dxc[2];
        if (dL_drot_canon) {
            if (k.rot6d) {
                float d6[6], dd[6];
#pragma unroll
                for (int i = 0; i < 6; i++) d6[i] = k.rot_canon[6 * (size_t)idx + i];
                sg_r6d2m_bwd(d6, dRc, dd);
#pragma unroll
                for (int i = 0; i < 6; i++) dL_drot_canon[6 * (size_t)idx + i] = dd[i] + (accumulate ? dL_drot_canon[6 * (size_t)idx + i] : 0.0f);
            } else {
#pragma unroll
                for (int i = 0; i < 9; i++) dL_drot_canon[9 * (size_t)idx + i] = dRc[i] + (accumulate ? dL_drot_canon[9 * (size_t)idx + i] : 0.0f);
            }
        }
        dL_dscales[3 * idx] = dsc[0]; dL_dscales[3 * idx + 1] = dsc[1]; dL_dscales[3 * idx + 2] = dsc[2];
        dL_dopacity[idx] = dop;
        dL_dmeans2D[3 * idx] = g2[0]; dL_dmeans2D[3 * idx + 1] = g2[1]; dL_dmeans2D[3 * idx + 2] = 0.0f;
    }
    // dL/dsh: coefficient-major planes [M][P][3], only the (D+1)^2 in use (SG_FLAG_SH_PLANAR: 12-byte stores, coalesced across
    // the lanes); or the reference's rows (every one of the M rows is written): through LDS as 16-B-per-lane coalesced stores when
    // M == 16
    if (c.flags & SG_FLAG_SH_PLANAR) {
        if (live) {
#pragma unroll
            for (int kq = 0; kq < nc; kq++)
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    float *d = dL_dsh + ((size_t)kq * P + idx) * 3 + ch;
                    *d = accumulate ? dsh[3 * kq + ch] + *d : dsh[3 * kq + ch];
                }
        }
    } else if (Mrows == 16) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if ((lane >> 5) == h) {
                float *row = sWw + (lane & 31) * SG_ROW_LDS;
#pragma unroll
                for (int i = 0; i < 48; i++) row[i] = i < nc * 3 ? dsh[i < nc * 3 ? i : 0] : 0.0f;
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
            sg_rows48_store(dL_dsh, g0 + 32 * h, P, lane, sWw, 32, accumulate != 0);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
        }
    } else if (live) {
        float *dsh_row = dL_dsh + (size_t)idx * Mrows * 3;
        if (accumulate) {
#pragma unroll
            for (int i = 0; i < nc * 3; i++) dsh_row[i] += dsh[i];
        } else {
#pragma unroll
            for (int i = 0; i < nc * 3; i++) dsh_row[i] = dsh[i];
            for (int i = nc * 3; i < Mrows * 3; i++) dsh_row[i] = 0.0f;
        }
    }
    // ---- dA[J x 16] += W^T[J x 64] . dT[64 x 16] on the matrix cores
    float *sdT = sTw;
#pragma unroll
    for (int i = 0; i < 12; i++) sdT[lane * SG_WSTRIDE + i] = dT[i];
#pragma unroll
    for (int i = 12; i < 16; i++) sdT[lane * SG_WSTRIDE + i] = 0.0f;
    const int nchunk = (k.J + 15) >> 4;
    // every wave owns one slab row (no cross-wave stage): [Jp x 16] dA partials + 3 dtransl partials
    float *out = slab + ((size_t)blockIdx.x * SG_SKIN_WAVES + wave) * slab_stride;
    const bool vec = (k.J & 3) == 0;
    float4 wv[4];
    if (vec) sg_w_fetch(k.lbs_w, k.J, P, g0, 0, lane, wv);
    for (int cch = 0; cch < nchunk; cch++) {
        if (vec) {
            sg_w_stash(wv, lane, sWw);
            if (cch + 1 < nchunk) sg_w_fetch(k.lbs_w, k.J, P, g0, cch + 1, lane, wv);
            __builtin_amdgcn_s_waitcnt(0xC07F);
        } else {
            sg_stage_w_chunk(k.lbs_w, k.J, P, g0, cch, lane, sWw);
            __builtin_amdgcn_s_waitcnt(0);
        }
        __builtin_amdgcn_wave_barrier();
        f32x4 acc = (f32x4){ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            float av = sWw[(4 * kk + (lane >> 4)) * SG_WSTRIDE + (lane & 15)];     // W^T[joint lane&15][gaussian]
            float bv = sdT[(4 * kk + (lane >> 4)) * SG_WSTRIDE + (lane & 15)];          // dT[gaussian][entry lane&15]
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
        }
        // acc[r] = dA[joint 16c + 4(lane>>4) + r][entry lane&15]
#pragma unroll
        for (int r = 0; r < 4; r++) out[(16 * cch + 4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[r];
        __builtin_amdgcn_wave_barrier();
    }
    // dtransl: wave sum
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float v = dtr[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) out[SG_JMAX * 16 + i] = v;
    }
}

// ---- K frames per launch (round 4) -----------------------------------------------------------------------------------------
// bt.K frames of the SAME canonical Gaussians: the wave walks the frames -- blend T with frame f's joint transforms, sum frame f's
// gradient records, chain rule, dA partials of frame f into frame f's slab row -- and sums the canonical-Gaussian gradients of the
// K frames in registers, in frame order: frame 0 assigns (ACC: adds to what the gradient buffer holds, read ONCE in front of the
// loop), frame f > 0 adds -- bit for bit what K single-frame calls leave behind when they share one gradient buffer (first call
// accumulate = 0 resp. 1, the others 1).  The gradient row -- 55 floats per Gaussian on the avatar, 48 of them the SH block -- is
// written ONCE per K frames instead of being read and rewritten by every frame.  dL_dmeans2D, dL_dA, dL_dtransl are per frame.
//
// What a frame costs here is dependent memory round trips, not arithmetic (6.3e6 VALU instructions per frame = 9 us of issue; the
// single-frame kernel takes 43 us, the first version of this loop 38 us per frame: A_f -> LDS, 4 weight chunks for T, slot / flags,
// 3 record chunks, the canonical inputs, 4 weight chunks again for W^T dT).  So, per wave:
//  * the skinning-weight tile (64 x J floats, the largest input: 4 J bytes per Gaussian) is read from HBM ONCE and stays in LDS
//    for all K frames and both contractions (row pitch J + 1: conflict-free for the [gaussian][joint] reads of W.A and the
//    [joint][gaussian] reads of W^T.dT alike);
//  * the canonical inputs (mean, scales, SH rows in use, 6-D rotation) are read once, in front of the loop;
//  * frame f + 1's joint transforms and (visible, record slot, clamp flags) are requested while frame f is worked on.
// LDS: 4 waves x (64 (J + 1) + 64 x 17) floats + 4 KB of joint transforms = 76 KB at J = 52: two workgroups per CU, as many
// waves per SIMD as the kernel's ~250 registers allow anyway.
// T rows 0..2 of this lane's Gaussian from the wave's RESIDENT weight tile sW [64][WP] (WP = J + 1) and the frame's sA.
// The same k-ordered fp32 MFMA chain as sg_skin_T (joints in ascending order, zero weights for the padding joints of the last
// chunk): the same T, bit for bit.
__device__ __forceinline__ void sg_skin_T_resident(const float *__restrict__ sW, int WP, int J, int lane, const float *__restrict__ sA,
                                                   float *__restrict__ sT, float T[12])
{
    f32x4 acc[4];
#pragma unroll
    for (int b = 0; b < 4; b++) acc[b] = (f32x4){ 0.0f, 0.0f, 0.0f, 0.0f };
    // k-steps of 4 joints; a step whose joints are all padding (J % 16 != 0: the tail of the last chunk) is skipped -- its weights are
    // zero and A is finite, so acc + 0 x b = acc exactly: the same T as sg_skin_T, bit for bit; the operands of a step are
    // requested together, in front of its four MFMAs (one LDS latency per step, not one per MFMA)
    const int nsteps = (J + 3) >> 2;
    for (int st = 0; st < nsteps; st++) {
        const int col = 4 * st + (lane >> 4);
        const float bv = sA[col * 16 + (lane & 15)];
        float av[4];
#pragma unroll
        for (int b = 0; b < 4; b++) av[b] = sW[(16 * b + (lane & 15)) * WP + (col < J ? col : 0)];
#pragma unroll
        for (int b = 0; b < 4; b++) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(col < J ? av[b] : 0.0f, bv, acc[b], 0, 0, 0);
    }
    if ((lane & 15) < 12) {
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) sT[(16 * b + 4 * (lane >> 4) + r) * 13 + (lane & 15)] = acc[b][r];
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 12; i++) T[i] = sT[lane * 13 + i];
    __builtin_amdgcn_wave_barrier();
}

// Record sums of ALL frames as a kernel of their own (grid: Gaussian blocks x K): a9 [K][P][12] (9 sums + padding: three 16-byte
// vectors per Gaussian).  Inside the frame loop below the sums were the largest item -- five dependent round trips (load -> LDS ->
// sum) per wave and frame, 98 of the loop kernel's 320 us at K = 8 (timing-only ablation builds, tools/ablate.py) -- because that
// kernel runs two waves per SIMD and nothing hides a round trip; here the same coalesced chunk loop runs at full occupancy (40
// registers) next to thousands of other waves, and the loop kernel reads 48 contiguous bytes per Gaussian and frame instead.  The
// order of the additions per Gaussian is sg_sum_records_coop's: the same sums, bit for bit.
__global__ void __launch_bounds__(256)
sg_record_sums_kernel(SgBatch bt, int P, const int32_t *__restrict__ radii0, SgGeom g0_, SgRec grec0, size_t cap,
                      const uint32_t *__restrict__ header0, const uint8_t *__restrict__ rec_valid, float4 *__restrict__ a9out)
{
    __shared__ float lds_all[4][SG_REC_CHUNK * 12];
    const int f = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (idx - lane >= P) return;
    const bool live = idx < P;
    const SgGeom g = sg_frame(g0_, (size_t)f * bt.geom);
    const SgRec grec = sg_frame(grec0, (size_t)f * bt.rec);
    const bool vis = live && radii0[(size_t)f * bt.P + idx] > 0 && sg_at(header0, (size_t)f * bt.bin)[1] == 0u;
    float4 rc = make_float4(0, 0, 0, 0);
    if (vis) { const uint2 sl = g.slot[idx]; rc.y = __uint_as_float(sl.x); rc.w = __uint_as_float(sl.y); }
    float a9[9];
    sg_sum_records_coop(grec, cap, vis, rc, lane, lds_all[wave], a9, sg_at(rec_valid, (size_t)f * bt.bin));
    if (live) {
        float4 *o = a9out + 3 * ((size_t)f * bt.P + idx);
        o[0] = make_float4(a9[0], a9[1], a9[2], a9[3]); o[1] = make_float4(a9[4], a9[5], a9[6], a9[7]);
        o[2] = make_float4(a9[8], 0.0f, 0.0f, 0.0f);
    }
}

void sg_launch_record_sums(const SgBatch &bt, int P, const int32_t *radii, SgGeom g, SgRec grec, size_t cap, const uint32_t *header,
                           const uint8_t *rec_valid, float4 *a9out, hipStream_t st)
{
    if (P <= 0) return;
    hipLaunchKernelGGL(sg_record_sums_kernel, dim3((P + 255) / 256, bt.K), dim3(256), 0, st, bt, P, radii, g, grec, cap, header,
                       rec_valid, a9out);
}

template <int D, bool ACC>
__global__ void __launch_bounds__(SG_SKIN_THREADS) __attribute__((amdgpu_waves_per_eu(D <= 1 ? 2 : 1, 2)))   // (D = 0: 255 VGPRs, no spills)
sg_skin_bwd_frames_kernel(SgCam c0, SgBatch bt, int P, SgSkin k0, const float *__restrict__ shs, const float *__restrict__ scales,
                          const int32_t *__restrict__ radii0, SgGeom g0_, SgRec grec0, size_t cap,
                          const uint32_t *__restrict__ header0, const float *__restrict__ dposed_xyz_in, const float *__restrict__ dposed_rotq_in,
                          float *__restrict__ dL_dxyz_canon, float *__restrict__ dL_drot_canon,
                          float *__restrict__ dL_dscales, float *__restrict__ dL_dopacity, float *__restrict__ dL_dsh,
                          float *__restrict__ dL_dmeans2D, float *__restrict__ slab, int slab_stride, size_t slab_frame,
                          const float4 *__restrict__ a9in)
{
    constexpr bool accumulate = ACC;
    extern __shared__ float sg_skin_lds[];                      // [sA 64 x 16][per wave: W tile 64 x WP | scratch 64 x 17]
    const int J = k0.J, WP = J + 1;
    const int wave_size = 64 * WP + 64 * SG_WSTRIDE;
    float *const sA = sg_skin_lds;
    const int wave_all = threadIdx.x >> 6, lane_all = threadIdx.x & 63;
    const int g0_all = (blockIdx.x * SG_SKIN_WAVES + wave_all) * 64;
    const int idx_all = g0_all + lane_all;
    const bool live = idx_all < P;
    const int Mrows = c0.M;
    constexpr int nc = (D + 1) * (D + 1);
    const bool have_in = dposed_xyz_in != nullptr || dposed_rotq_in != nullptr;
    const int nchunk = (J + 15) >> 4;
    // ---- once per K frames: the wave's weight tile into LDS (coalesced: the tile's 64 rows are one contiguous range)
    {
        float *const sW = sg_skin_lds + SG_JMAX * 16 + wave_all * wave_size;
        const int rows = P - g0_all < 64 ? (P - g0_all > 0 ? P - g0_all : 0) : 64;
        const int total = rows * J;
        const float *src = k0.lbs_w + (size_t)g0_all * J;
        if (((J & 3) == 0)) {
            // every load of the tile is issued before the first LDS write: ONE memory round trip (64 x 64 floats = 16 float4 per lane)
            float4 v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int e = 4 * (64 * u + lane_all);
                v[u] = e < total ? *(const float4 *)(src + e) : make_float4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int e = 4 * (64 * u + lane_all);
                if (e < total) {
                    const int r = e / J, cc = e - r * J;              // (J % 4 == 0: the four values share a row)
                    float *d = sW + r * WP + cc;
                    d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
                }
            }
        } else {
            for (int e = lane_all; e < total; e += 64) { const int r = e / J; sW[r * WP + (e - r * J)] = src[e]; }
        }
        for (int e = total + lane_all; e < 64 * J; e += 64) { const int r = e / J; sW[r * WP + (e - r * J)] = 0.0f; }   // rows beyond P
    }
    // ---- once per K frames: the canonical inputs of this lane's Gaussian
    float cx[3] = { 0, 0, 0 }, cs[3] = { 0, 0, 0 }, cRc[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, d6[6] = { 0, 0, 0, 0, 0, 0 };
    if (live) {
        cx[0] = k0.xyz_canon[3 * idx_all]; cx[1] = k0.xyz_canon[3 * idx_all + 1]; cx[2] = k0.xyz_canon[3 * idx_all + 2];
        cs[0] = scales[3 * idx_all]; cs[1] = scales[3 * idx_all + 1]; cs[2] = scales[3 * idx_all + 2];
        if (k0.rot_canon) {
            if (k0.rot6d) {
#pragma unroll
                for (int i = 0; i < 6; i++) d6[i] = k0.rot_canon[6 * (size_t)idx_all + i];
                sg_r6d2m(d6, cRc);
            } else {
#pragma unroll
                for (int i = 0; i < 9; i++) cRc[i] = k0.rot_canon[9 * (size_t)idx_all + i];
            }
        }
    }
    const float csc = k0.smpl_scale ? k0.smpl_scale[0] : 1.0f;
    // ---- sums over the frames; ACC: they start from what the gradient buffer holds
    float dxc[3] = { 0, 0, 0 }, dRc[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, dsc[3] = { 0, 0, 0 }, dop = 0;
    float dsh[nc * 3];
#pragma unroll
    for (int i = 0; i < nc * 3; i++) dsh[i] = 0.0f;
    if (accumulate && live) {
#pragma unroll
        for (int i = 0; i < 3; i++) { dxc[i] = dL_dxyz_canon[3 * idx_all + i]; dsc[i] = dL_dscales[3 * idx_all + i]; }
        dop = dL_dopacity[idx_all];
        if (dL_drot_canon) {
            const int rw = k0.rot6d ? 6 : 9;
            for (int i = 0; i < rw; i++) dRc[i] = dL_drot_canon[(size_t)rw * idx_all + i];
        }
        const bool planar = c0.flags & SG_FLAG_SH_PLANAR;
#pragma unroll
        for (int i = 0; i < nc * 3; i++)
            dsh[i] = planar ? dL_dsh[((size_t)(i / 3) * P + idx_all) * 3 + i % 3] : dL_dsh[(size_t)idx_all * Mrows * 3 + i];
    }
    // frame 0's joint transforms; (visible, slot, flags) of frame 0
    for (int i = threadIdx.x; i < SG_JMAX * 16; i += SG_SKIN_THREADS) sA[i] = i < J * 16 ? k0.A[i] : 0.0f;
    bool vis_n = live && radii0[idx_all] > 0 && header0[1] == 0u;
    uint32_t fl_n = 0u;
    if (vis_n) fl_n = g0_.flags[idx_all];
    __syncthreads();
#pragma unroll 1
    for (int f = 0; f < bt.K; f++) {
        // lane / wave / Gaussian index are made opaque per trip: hipcc otherwise hoists everything of the loop body that depends on
        // them only (LDS addresses, shuffle lane indices, per-frame array addresses) in front of the loop and keeps it live across
        // it -- 330 VGPRs instead of ~250, one wave per SIMD instead of two
        int idx = idx_all, lane = lane_all, wave = wave_all;
        asm volatile("" : "+v"(idx), "+v"(lane), "+v"(wave));
        float *const sW = sg_skin_lds + SG_JMAX * 16 + wave * wave_size, *const sTw = sW + 64 * WP;
        const SgCam c = sg_frame(c0, f, bt.cam_stride);
        const float *transl = k0.transl ? k0.transl + (size_t)f * bt.transl_stride : nullptr;
        // next frame's joint transforms: requested now, written to LDS when every wave is done with this frame's
        float4 an = make_float4(0, 0, 0, 0);
        const bool more = f + 1 < bt.K;
        if (more) {
            const int i4 = 4 * (int)threadIdx.x;                  // SG_JMAX * 16 = 1024 floats = one float4 per thread
            if (i4 < J * 16) an = *(const float4 *)(k0.A + (size_t)(f + 1) * J * 16 + i4);
        }
        float T[12];
        sg_skin_T_resident(sW, WP, J, lane, sA, sTw, T);
        const bool vis = vis_n;
        const uint32_t flags = fl_n;
        if (more) {
            const SgGeom gn = sg_frame(g0_, (size_t)(f + 1) * bt.geom);
            vis_n = live && radii0[(size_t)(f + 1) * bt.P + idx] > 0 && sg_at(header0, (size_t)(f + 1) * bt.bin)[1] == 0u;
            fl_n = 0u;
            if (vis_n) fl_n = gn.flags[idx];
        }
        // this frame's record sums (sg_record_sums_kernel): 48 contiguous bytes per Gaussian
        float a9[9];
        {
            const float4 *src = a9in + 3 * ((size_t)f * bt.P + idx);
            float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0;
            if (vis) { r0 = src[0]; r1 = src[1]; r2 = src[2]; }
            a9[0] = r0.x; a9[1] = r0.y; a9[2] = r0.z; a9[3] = r0.w; a9[4] = r1.x; a9[5] = r1.y; a9[6] = r1.z; a9[7] = r1.w; a9[8] = r2.x;
        }
        float dT[12];
#pragma unroll
        for (int i = 0; i < 12; i++) dT[i] = 0.0f;
        float dtr[3] = { 0, 0, 0 };
        if (live) {
            float fxc[3] = { 0, 0, 0 }, fRc[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, fsc[3] = { 0, 0, 0 }, fop = 0, g2[2] = { 0, 0 };
            if (vis || have_in) {
                // sings_hybrid.py:400-419 for this Gaussian (sg_pose_gaussian without the loads and without ext_tfs)
                float pp[3], s3[3], Rdef[9], q[4];
#pragma unroll
                for (int i = 0; i < 3; i++) pp[i] = T[4 * i] * cx[0] + T[4 * i + 1] * cx[1] + T[4 * i + 2] * cx[2] + T[4 * i + 3];
                s3[0] = cs[0]; s3[1] = cs[1]; s3[2] = cs[2];
                if (k0.smpl_scale) {
#pragma unroll
                    for (int i = 0; i < 3; i++) { pp[i] = pp[i] * csc; s3[i] = s3[i] * csc; }
                }
                if (transl) {
#pragma unroll
                    for (int i = 0; i < 3; i++) pp[i] = pp[i] + transl[i];
                }
                if (k0.rot_canon) {
#pragma unroll
                    for (int i = 0; i < 3; i++)
#pragma unroll
                        for (int j = 0; j < 3; j++)
                            Rdef[3 * i + j] = T[4 * i] * cRc[j] + T[4 * i + 1] * cRc[3 + j] + T[4 * i + 2] * cRc[6 + j];
                } else {
#pragma unroll
                    for (int i = 0; i < 3; i++)
#pragma unroll
                        for (int j = 0; j < 3; j++) Rdef[3 * i + j] = T[4 * i + j];
                }
                int best; float qab;
                sg_m2q(Rdef, q, best, qab);
                SgGaussGrad G;
#pragma unroll
                for (int i = 0; i < 3; i++) { G.dmean[i] = 0; G.dsc[i] = 0; G.dcol[i] = 0; }
#pragma unroll
                for (int i = 0; i < 4; i++) G.drot[i] = 0;
                G.g2[0] = G.g2[1] = 0; G.dop = 0;
                if (vis)
                    sg_project_bwd<D>(c, pp, s3, q, nullptr, shs + (size_t)idx * Mrows * 3, flags, a9, true, dsh, G, !accumulate && f == 0);
                if (dposed_xyz_in) {
#pragma unroll
                    for (int i = 0; i < 3; i++) G.dmean[i] += dposed_xyz_in[3 * ((size_t)f * bt.P + idx) + i];
                }
                if (dposed_rotq_in) {
#pragma unroll
                    for (int i = 0; i < 4; i++) G.drot[i] += dposed_rotq_in[4 * ((size_t)f * bt.P + idx) + i];
                }
                g2[0] = G.g2[0]; g2[1] = G.g2[1]; fop = G.dop;
#pragma unroll
                for (int i = 0; i < 3; i++) { fsc[i] = G.dsc[i] * csc; dtr[i] = G.dmean[i]; }
                float dps[3] = { G.dmean[0] * csc, G.dmean[1] * csc, G.dmean[2] * csc };
                float dRd[9];
                sg_m2q_bwd(q, best, qab, G.drot, dRd);
#pragma unroll
                for (int i = 0; i < 3; i++) {
#pragma unroll
                    for (int kx = 0; kx < 3; kx++)
                        dT[4 * i + kx] = dps[i] * cx[kx] + dRd[3 * i] * cRc[3 * kx] + dRd[3 * i + 1] * cRc[3 * kx + 1] + dRd[3 * i + 2] * cRc[3 * kx + 2];
                    dT[4 * i + 3] = dps[i];
                }
#pragma unroll
                for (int kx = 0; kx < 3; kx++) {
                    fxc[kx] = T[kx] * dps[0] + T[4 + kx] * dps[1] + T[8 + kx] * dps[2];
#pragma unroll
                    for (int j = 0; j < 3; j++) fRc[3 * kx + j] = T[kx] * dRd[j] + T[4 + kx] * dRd[3 + j] + T[8 + kx] * dRd[6 + j];
                }
            }
            // the decoder's 6-D rotations: every frame's dL/dR_canon goes through the Gram-Schmidt backward on its own, as in a
            // single-frame call (the map is linear in dR, but J (a + b) and J a + J b differ in the last bit)
            if (dL_drot_canon && k0.rot6d) {
                float dd[6];
                sg_r6d2m_bwd(d6, fRc, dd);
#pragma unroll
                for (int i = 0; i < 6; i++) fRc[i] = dd[i];
            }
            if (!accumulate && f == 0) {
#pragma unroll
                for (int i = 0; i < 3; i++) { dxc[i] = fxc[i]; dsc[i] = fsc[i]; }
#pragma unroll
                for (int i = 0; i < 9; i++) dRc[i] = fRc[i];
                dop = fop;
            } else {
                // (operand order of the single-frame accumulate: new + old)
#pragma unroll
                for (int i = 0; i < 3; i++) { dxc[i] = fxc[i] + dxc[i]; dsc[i] = fsc[i] + dsc[i]; }
#pragma unroll
                for (int i = 0; i < 9; i++) dRc[i] = fRc[i] + dRc[i];
                dop = fop + dop;
            }
            float *m2 = dL_dmeans2D + 3 * ((size_t)f * bt.P + idx);
            m2[0] = g2[0]; m2[1] = g2[1]; m2[2] = 0.0f;
        }
        // ---- dA[J x 16] = W^T[J x 64] . dT[64 x 16] on the matrix cores, W from the resident tile
        float *sdT = sTw;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 12; i++) sdT[lane * SG_WSTRIDE + i] = dT[i];
#pragma unroll
        for (int i = 12; i < 16; i++) sdT[lane * SG_WSTRIDE + i] = 0.0f;
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        float *out = slab + (size_t)f * slab_frame + ((size_t)blockIdx.x * SG_SKIN_WAVES + wave) * slab_stride;
        for (int cch = 0; cch < nchunk; cch++) {
            f32x4 acc = (f32x4){ 0.0f, 0.0f, 0.0f, 0.0f };
            // (joints >= J of the last chunk: the column index is clamped -- those rows of dA are written with whatever comes out
            //  and never read: sg_skin_reduce2_kernel sums J x 16 columns)
            const int col = 16 * cch + (lane & 15) < J ? 16 * cch + (lane & 15) : J - 1;
#pragma unroll
            for (int k4 = 0; k4 < 4; k4++) {
                float av[4], bv[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int kk = 4 * k4 + u;
                    av[u] = sW[(4 * kk + (lane >> 4)) * WP + col];                          // W^T[joint][gaussian 4 kk + (lane >> 4)]
                    bv[u] = sdT[(4 * kk + (lane >> 4)) * SG_WSTRIDE + (lane & 15)];          // dT[gaussian][entry lane & 15]
                }
#pragma unroll
                for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) out[(16 * cch + 4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[r];
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
            float v = dtr[i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) out[SG_JMAX * 16 + i] = v;
        }
        if (more) {
            __syncthreads();                                     // every wave is done with this frame's joint transforms
            *(float4 *)(sA + 4 * (int)threadIdx.x) = an;
            __syncthreads();
        }
    }
    // ---- the canonical-Gaussian gradients, once for the K frames (ACC: the sums already include what the buffer held)
    const int idx = idx_all, lane = lane_all, g0 = g0_all;
    float *const sWw = sg_skin_lds + SG_JMAX * 16 + wave_all * wave_size;      // (the weight tile is dead: staging for the SH rows)
    if (live) {
        dL_dxyz_canon[3 * idx] = dxc[0]; dL_dxyz_canon[3 * idx + 1] = dxc[1]; dL_dxyz_canon[3 * idx + 2] = dxc[2];
        if (dL_drot_canon) {
            const int rw = k0.rot6d ? 6 : 9;
            for (int i = 0; i < rw; i++) dL_drot_canon[(size_t)rw * idx + i] = dRc[i];
        }
        dL_dscales[3 * idx] = dsc[0]; dL_dscales[3 * idx + 1] = dsc[1]; dL_dscales[3 * idx + 2] = dsc[2];
        dL_dopacity[idx] = dop;
    }
    // dL/dsh rows: every one of the M rows is written (through LDS as 16-B-per-lane coalesced stores when M == 16); ACC: only the
    // (D+1)^2 rows in use are rewritten -- the others keep what the step's first call wrote
    if (c0.flags & SG_FLAG_SH_PLANAR) {                          // coefficient-major planes, only those in use (see sg_skin_bwd_kernel)
        if (live) {
#pragma unroll
            for (int kq = 0; kq < nc; kq++)
#pragma unroll
                for (int ch = 0; ch < 3; ch++) dL_dsh[((size_t)kq * P + idx) * 3 + ch] = dsh[3 * kq + ch];
        }
    } else if (Mrows == 16 && (!accumulate || nc == 16) && 64 * WP >= 32 * SG_ROW_LDS) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if ((lane >> 5) == h) {
                float *row = sWw + (lane & 31) * SG_ROW_LDS;
#pragma unroll
                for (int i = 0; i < 48; i++) row[i] = i < nc * 3 ? dsh[i < nc * 3 ? i : 0] : 0.0f;
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
            sg_rows48_store(dL_dsh, g0 + 32 * h, P, lane, sWw, 32, false);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
        }
    } else if (live) {
        float *dsh_row = dL_dsh + (size_t)idx * Mrows * 3;
#pragma unroll
        for (int i = 0; i < nc * 3; i++) dsh_row[i] = dsh[i];
        if (!accumulate)
            for (int i = nc * 3; i < Mrows * 3; i++) dsh_row[i] = 0.0f;
    }
}

// sums the per-wave slab rows in a fixed order: dL_dA [J,16] and dL_dtransl [3].  Two passes so that the
// ~10 MB of slabs are read by thousands of waves with 256-B coalesced rows (a single 53-block pass took 44 us):
//  pass 1: grid (column blocks of 64, SG_RED_GROUPS/4); wave = row group rg sums rows rg, rg + G, ... into part[rg]
//  pass 2: one thread per column sums the G partial rows.
#define SG_RED_GROUPS 64        // (32 / 64 / 128 groups: pass 1 + pass 2 = 8.5 + 4.7 / 5.6 + 4.9 / 4.9 + 6.8 us on the avatar frame)
__global__ void __launch_bounds__(256)
sg_skin_reduce1_kernel(const float *__restrict__ slab, int nrows, int slab_stride, float *__restrict__ part, size_t slab_frame)
{
    slab += (size_t)blockIdx.z * slab_frame;                   // frame blockIdx.z: its slab rows -> its SG_RED_GROUPS partial rows
    part += (size_t)blockIdx.z * SG_RED_GROUPS * slab_stride;
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rg = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (col >= slab_stride) return;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int b = rg;
    for (; b + 3 * SG_RED_GROUPS < nrows; b += 4 * SG_RED_GROUPS) {
        s0 += slab[(size_t)b * slab_stride + col];
        s1 += slab[(size_t)(b + SG_RED_GROUPS) * slab_stride + col];
        s2 += slab[(size_t)(b + 2 * SG_RED_GROUPS) * slab_stride + col];
        s3 += slab[(size_t)(b + 3 * SG_RED_GROUPS) * slab_stride + col];
    }
    for (; b < nrows; b += SG_RED_GROUPS) s0 += slab[(size_t)b * slab_stride + col];
    part[(size_t)rg * slab_stride + col] = (s0 + s1) + (s2 + s3);
}

__global__ void __launch_bounds__(64)
sg_skin_reduce2_kernel(const float *__restrict__ part, int slab_stride, int J, float *__restrict__ dL_dA,
                       float *__restrict__ dL_dtransl)
{
    part += (size_t)blockIdx.z * SG_RED_GROUPS * slab_stride;  // frame blockIdx.z -> dL_dA [K,J,16], dL_dtransl [K,3]
    dL_dA += (size_t)blockIdx.z * J * 16;
    if (dL_dtransl) dL_dtransl += 3 * (size_t)blockIdx.z;
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int nA = J * 16;
    if (i >= nA + 3) return;
    const int col = i < nA ? i : SG_JMAX * 16 + (i - nA);
    float t = 0.0f;
#pragma unroll 8
    for (int r = 0; r < SG_RED_GROUPS; r++) t += part[(size_t)r * slab_stride + col];
    if (i < nA) dL_dA[i] = t;
    else if (dL_dtransl) dL_dtransl[i - nA] = t;
}

// ---- stand-alone lbs_extra (sings/rec/utils/body_model/lbs.py:59-74) ---------------------------------------------
// For callers that keep SinGS.forward as it is (sings_hybrid.py:400-406) instead of the fused path: T[N,4,4] = W . A and
// verts = (T [v;1])[:3] in one pass (the reference: expand + matmul + cat + matmul + slice, T and v_homo through HBM
// twice), and the transpose for autograd.  Same MFMA contraction as above, all 16 columns kept.
#define SG_T16STRIDE 17
__device__ __forceinline__ void sg_skin_T16(const float *__restrict__ W, int J, int P, int g0, int lane,
                                            const float *__restrict__ sA, float *__restrict__ sW,
                                            float *__restrict__ sT, float T[16])
{
    f32x4 acc[4];
#pragma unroll
    for (int b = 0; b < 4; b++) acc[b] = (f32x4){ 0.0f, 0.0f, 0.0f, 0.0f };
    const int nchunk = (J + 15) >> 4;
    for (int c = 0; c < nchunk; c++) {
        sg_stage_w_chunk(W, J, P, g0, c, lane, sW);
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            float bv = sA[(c * 16 + 4 * kk + (lane >> 4)) * 16 + (lane & 15)];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                float av = sW[(16 * b + (lane & 15)) * SG_WSTRIDE + 4 * kk + (lane >> 4)];
                acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[b], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) sT[(16 * b + 4 * (lane >> 4) + r) * SG_T16STRIDE + (lane & 15)] = acc[b][r];
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 16; i++) T[i] = sT[lane * SG_T16STRIDE + i];
    __builtin_amdgcn_wave_barrier();
}

__global__ void __launch_bounds__(SG_SKIN_THREADS)
sg_lbs_fwd_kernel(int P, int J, const float *__restrict__ W, const float *__restrict__ A, const float *__restrict__ v,
                  float *__restrict__ T_out, float *__restrict__ verts)
{
    __shared__ float sA[SG_JMAX * 16];
    __shared__ float sW[SG_SKIN_WAVES][64 * SG_WSTRIDE];
    __shared__ float sT[SG_SKIN_WAVES][64 * SG_T16STRIDE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g0 = (blockIdx.x * SG_SKIN_WAVES + wave) * 64, idx = g0 + lane;
    const int Jp = ((J + 15) >> 4) << 4;
    for (int i = threadIdx.x; i < Jp * 16; i += blockDim.x) sA[i] = i < J * 16 ? A[i] : 0.0f;
    __syncthreads();
    float T[16];
    sg_skin_T16(W, J, P, g0, lane, sA, sW[wave], sT[wave], T);
    if (idx >= P) return;
    const float x = v[3 * idx], y = v[3 * idx + 1], z = v[3 * idx + 2];
    if (T_out) {
        float4 *o = (float4 *)(T_out + (size_t)idx * 16);
#pragma unroll
        for (int r = 0; r < 4; r++) o[r] = make_float4(T[4 * r], T[4 * r + 1], T[4 * r + 2], T[4 * r + 3]);
    }
    // (T @ [v;1]): four products summed left to right, as torch.matmul's K = 4 reduction
#pragma unroll
    for (int r = 0; r < 3; r++) verts[3 * idx + r] = ((T[4 * r] * x + T[4 * r + 1] * y) + T[4 * r + 2] * z) + T[4 * r + 3];
}

// dT_in [P,16] and / or dverts_in [P,3] (either may be NULL) -> dv [P,3], per-wave dA slabs (summed by the reduce kernels)
__global__ void __launch_bounds__(SG_SKIN_THREADS)
sg_lbs_bwd_kernel(int P, int J, const float *__restrict__ W, const float *__restrict__ A, const float *__restrict__ v,
                  const float *__restrict__ dT_in, const float *__restrict__ dverts_in, float *__restrict__ dv,
                  float *__restrict__ slab, int slab_stride)
{
    __shared__ float sA[SG_JMAX * 16];
    __shared__ float sW[SG_SKIN_WAVES][64 * SG_WSTRIDE];
    __shared__ float sT[SG_SKIN_WAVES][64 * SG_T16STRIDE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g0 = (blockIdx.x * SG_SKIN_WAVES + wave) * 64, idx = g0 + lane;
    const int Jp = ((J + 15) >> 4) << 4;
    for (int i = threadIdx.x; i < Jp * 16; i += blockDim.x) sA[i] = i < J * 16 ? A[i] : 0.0f;
    __syncthreads();
    float T[16], dT[16];
    sg_skin_T16(W, J, P, g0, lane, sA, sW[wave], sT[wave], T);
    const bool live = idx < P;
#pragma unroll
    for (int i = 0; i < 16; i++) dT[i] = (live && dT_in) ? dT_in[(size_t)idx * 16 + i] : 0.0f;
    if (live && dverts_in) {
        const float g[3] = { dverts_in[3 * idx], dverts_in[3 * idx + 1], dverts_in[3 * idx + 2] };
        const float vh[4] = { v[3 * idx], v[3 * idx + 1], v[3 * idx + 2], 1.0f };
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int cc = 0; cc < 4; cc++) dT[4 * r + cc] += g[r] * vh[cc];
        if (dv) {
#pragma unroll
            for (int cc = 0; cc < 3; cc++) dv[3 * idx + cc] = (T[cc] * g[0] + T[4 + cc] * g[1]) + T[8 + cc] * g[2];
        }
    } else if (live && dv) {
        dv[3 * idx] = 0.0f; dv[3 * idx + 1] = 0.0f; dv[3 * idx + 2] = 0.0f;
    }
    float *sdT = sT[wave];
#pragma unroll
    for (int i = 0; i < 16; i++) sdT[lane * SG_WSTRIDE + i] = dT[i];
    float *out = slab + ((size_t)blockIdx.x * SG_SKIN_WAVES + wave) * slab_stride;
    const int nchunk = (J + 15) >> 4;
    for (int cch = 0; cch < nchunk; cch++) {
        sg_stage_w_chunk(W, J, P, g0, cch, lane, sW[wave]);
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        f32x4 acc = (f32x4){ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            float av = sW[wave][(4 * kk + (lane >> 4)) * SG_WSTRIDE + (lane & 15)];
            float bv = sdT[(4 * kk + (lane >> 4)) * SG_WSTRIDE + (lane & 15)];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) out[(16 * cch + 4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[r];
        __builtin_amdgcn_wave_barrier();
    }
    for (int cch = nchunk; cch < SG_JMAX / 16; cch++)                      // rows the reduce kernels also read
#pragma unroll
        for (int r = 0; r < 4; r++) out[(16 * cch + 4 * (lane >> 4) + r) * 16 + (lane & 15)] = 0.0f;
    if (lane < 4) out[SG_JMAX * 16 + lane] = 0.0f;
}

// ---- stand-alone matrix_to_quaternion (rotations.py:98-149) -------------------------------------------------
// One lane per matrix (the reference: ~25 element-wise / indexing kernels each way).  Same operation order as the torch
// expression -> bit-identical quaternions (golden G1).
__global__ void __launch_bounds__(256)
sg_m2q_fwd_kernel(int N, const float *__restrict__ m, float *__restrict__ q)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float mm[9], qq[4], qab;
    int best;
#pragma unroll
    for (int k = 0; k < 9; k++) mm[k] = m[9 * (size_t)i + k];
    sg_m2q(mm, qq, best, qab);
    *(float4 *)(q + 4 * (size_t)i) = make_float4(qq[0], qq[1], qq[2], qq[3]);
}
__global__ void __launch_bounds__(256)
sg_m2q_bwd_kernel(int N, const float *__restrict__ m, const float *__restrict__ dq, float *__restrict__ dm)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float mm[9], qq[4], qab, g[4], d[9];
    int best;
#pragma unroll
    for (int k = 0; k < 9; k++) mm[k] = m[9 * (size_t)i + k];
    sg_m2q(mm, qq, best, qab);
    const float4 gq = *(const float4 *)(dq + 4 * (size_t)i);
    g[0] = gq.x; g[1] = gq.y; g[2] = gq.z; g[3] = gq.w;
    sg_m2q_bwd(qq, best, qab, g, d);
#pragma unroll
    for (int k = 0; k < 9; k++) dm[9 * (size_t)i + k] = d[k];
}
void sg_launch_m2q(int N, const float *m, const float *dq, float *out, hipStream_t st)
{
    if (N <= 0) return;
    if (dq) hipLaunchKernelGGL(sg_m2q_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, st, N, m, dq, out);
    else hipLaunchKernelGGL(sg_m2q_fwd_kernel, dim3((N + 255) / 256), dim3(256), 0, st, N, m, out);
}

// ---- SMPL(-H) kinematic chain (sings/rec/utils/body_model/smpl.py:415-513) ------------------------------------
// pose [B,J,3] axis-angle + rest joints [J,3] + parents -> A [B,J,16] = G - pad(G [J;0]) (x post[j] if given: the
// per-joint inv(A_t2cano) of sings_hybrid.py:398-399).  torch runs this as ~100 tiny launches per call (Rodrigues, J - 1
// dependent 4x4 products, the correction) and three times that for autograd: launch latency, ~1 ms per frame.  Here:
// one wave per frame, lane = joint; the chain is walked in joint order (parents[i] < i) through LDS.
struct SgRod { float R[9], th, d[3], s, c; };
__device__ __forceinline__ void sg_rodrigues(const float v[3], SgRod &o)
{
    // batch_rodrigues (smpl.py:415-446): angle = |v + 1e-8|, axis = v / angle
    const float a0 = v[0] + 1e-8f, a1 = v[1] + 1e-8f, a2 = v[2] + 1e-8f;
    o.th = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
    o.d[0] = v[0] / o.th; o.d[1] = v[1] / o.th; o.d[2] = v[2] / o.th;
    o.s = sinf(o.th); o.c = cosf(o.th);
    const float K[9] = { 0, -o.d[2], o.d[1], o.d[2], 0, -o.d[0], -o.d[1], o.d[0], 0 };
    const float oc = 1.0f - o.c;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int cc = 0; cc < 3; cc++) {
            const float kk = K[3 * r] * K[cc] + K[3 * r + 1] * K[3 + cc] + K[3 * r + 2] * K[6 + cc];
            o.R[3 * r + cc] = (r == cc ? 1.0f : 0.0f) + o.s * K[3 * r + cc] + oc * kk;
        }
}
// dL/dR [9] -> dL/dv [3]
__device__ __forceinline__ void sg_rodrigues_bwd(const float v[3], const SgRod &o, const float M[9], float dv[3])
{
    const float K[9] = { 0, -o.d[2], o.d[1], o.d[2], 0, -o.d[0], -o.d[1], o.d[0], 0 };
    float K2[9], mk = 0.0f, mk2 = 0.0f;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int cc = 0; cc < 3; cc++) {
            K2[3 * r + cc] = K[3 * r] * K[cc] + K[3 * r + 1] * K[3 + cc] + K[3 * r + 2] * K[6 + cc];
            mk += M[3 * r + cc] * K[3 * r + cc]; mk2 += M[3 * r + cc] * K2[3 * r + cc];
        }
    const float oc = 1.0f - o.c;
    float dK[9];                                                  // s M + (1 - c) (M K^T + K^T M)
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int cc = 0; cc < 3; cc++) {
            float a = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; k++) a += M[3 * r + k] * K[3 * cc + k] + K[3 * k + r] * M[3 * k + cc];
            dK[3 * r + cc] = o.s * M[3 * r + cc] + oc * a;
        }
    const float dd[3] = { dK[7] - dK[5], dK[2] - dK[6], dK[3] - dK[1] };
    float dth = o.c * mk + o.s * mk2;
    const float ith = 1.0f / o.th;
#pragma unroll
    for (int k = 0; k < 3; k++) dth -= dd[k] * v[k] * ith * ith;
#pragma unroll
    for (int k = 0; k < 3; k++) dv[k] = dd[k] * ith + dth * (v[k] + 1e-8f) * ith;
}
// C[3x4] = A[3x4] o B[3x4] as rigid 4x4 products (bottom rows 0 0 0 1)
__device__ __forceinline__ void sg_mul34(const float *A, const float *B, float *Cm)
{
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int cc = 0; cc < 4; cc++)
            Cm[4 * r + cc] = (A[4 * r] * B[cc] + A[4 * r + 1] * B[4 + cc]) + A[4 * r + 2] * B[8 + cc] + (cc == 3 ? A[4 * r + 3] : 0.0f);
    }
}
__global__ void __launch_bounds__(64)
sg_joint_transforms_fwd_kernel(int J, const float *__restrict__ pose, const float *__restrict__ joints, const int *__restrict__ parents,
                               const float *__restrict__ post, float *__restrict__ A_out)
{
    __shared__ float sG[64][13];
    const int b = blockIdx.x, j = threadIdx.x;
    const bool on = j < J;
    float T[12], G[12];
    const int par = on ? parents[j] : -1;
    float Jj[3] = { 0, 0, 0 };
    if (on) {
        const float v[3] = { pose[((size_t)b * J + j) * 3], pose[((size_t)b * J + j) * 3 + 1], pose[((size_t)b * J + j) * 3 + 2] };
        SgRod ro;
        sg_rodrigues(v, ro);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            Jj[k] = joints[3 * j + k];
            T[4 * k] = ro.R[3 * k]; T[4 * k + 1] = ro.R[3 * k + 1]; T[4 * k + 2] = ro.R[3 * k + 2];
            T[4 * k + 3] = Jj[k] - (par >= 0 ? joints[3 * par + k] : 0.0f);
        }
    }
    for (int i = 0; i < J; i++) {
        if (j == i) {
            if (par < 0) {
#pragma unroll
                for (int k = 0; k < 12; k++) G[k] = T[k];
            } else {
                float P[12];
#pragma unroll
                for (int k = 0; k < 12; k++) P[k] = sG[par][k];
                sg_mul34(P, T, G);
            }
#pragma unroll
            for (int k = 0; k < 12; k++) sG[j][k] = G[k];
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    }
    if (!on) return;
    // A = G - pad(G [J;0]): same rotation, translation t - R J
#pragma unroll
    for (int r = 0; r < 3; r++) G[4 * r + 3] -= (G[4 * r] * Jj[0] + G[4 * r + 1] * Jj[1]) + G[4 * r + 2] * Jj[2];
    float *o = A_out + ((size_t)b * J + j) * 16;
    if (post) {
        const float *Pm = post + (size_t)j * 16;                  // full 4x4: A (bottom row 0 0 0 1) @ post
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int cc = 0; cc < 4; cc++)
                o[4 * r + cc] = ((G[4 * r] * Pm[cc] + G[4 * r + 1] * Pm[4 + cc]) + G[4 * r + 2] * Pm[8 + cc]) + G[4 * r + 3] * Pm[12 + cc];
#pragma unroll
        for (int cc = 0; cc < 4; cc++) o[12 + cc] = Pm[12 + cc];
    } else {
#pragma unroll
        for (int k = 0; k < 12; k++) o[k] = G[k];
        o[12] = 0.0f; o[13] = 0.0f; o[14] = 0.0f; o[15] = 1.0f;
    }
}
// dA [B,J,16] -> dpose [B,J,3], djoints [B,J,3] (per frame; the caller sums over B)
__global__ void __launch_bounds__(64)
sg_joint_transforms_bwd_kernel(int J, const float *__restrict__ pose, const float *__restrict__ joints, const int *__restrict__ parents,
                               const float *__restrict__ post, const float *__restrict__ dA, float *__restrict__ dpose,
                               float *__restrict__ djoints)
{
    __shared__ float sG[64][13], sD[64][13], sJ[64][4];
    const int b = blockIdx.x, j = threadIdx.x;
    const bool on = j < J;
    float T[12], G[12], dG[12];
    const int par = on ? parents[j] : -1;
    float Jj[3] = { 0, 0, 0 }, v[3] = { 0, 0, 0 };
    SgRod ro;
    if (on) {
#pragma unroll
        for (int k = 0; k < 3; k++) v[k] = pose[((size_t)b * J + j) * 3 + k];
        sg_rodrigues(v, ro);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            Jj[k] = joints[3 * j + k];
            T[4 * k] = ro.R[3 * k]; T[4 * k + 1] = ro.R[3 * k + 1]; T[4 * k + 2] = ro.R[3 * k + 2];
            T[4 * k + 3] = Jj[k] - (par >= 0 ? joints[3 * par + k] : 0.0f);
        }
    }
    for (int i = 0; i < J; i++) {                                 // forward chain again (G into LDS)
        if (j == i) {
            if (par < 0) {
#pragma unroll
                for (int k = 0; k < 12; k++) G[k] = T[k];
            } else {
                float P[12];
#pragma unroll
                for (int k = 0; k < 12; k++) P[k] = sG[par][k];
                sg_mul34(P, T, G);
            }
#pragma unroll
            for (int k = 0; k < 12; k++) sG[j][k] = G[k];
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    }
    // dA (after post) -> dA of the un-multiplied transform (top three rows) -> dG, direct dJ
    float dJ[3] = { 0, 0, 0 };
#pragma unroll
    for (int k = 0; k < 12; k++) dG[k] = 0.0f;
    if (on) {
        const float *g = dA + ((size_t)b * J + j) * 16;
        float dAp[12];
        if (post) {
            const float *Pm = post + (size_t)j * 16;              // d(A) = dOut @ post^T, rows 0..2
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int k = 0; k < 4; k++)
                    dAp[4 * r + k] = ((g[4 * r] * Pm[4 * k] + g[4 * r + 1] * Pm[4 * k + 1]) + g[4 * r + 2] * Pm[4 * k + 2]) + g[4 * r + 3] * Pm[4 * k + 3];
        } else {
#pragma unroll
            for (int k = 0; k < 12; k++) dAp[k] = g[k];
        }
#pragma unroll
        for (int r = 0; r < 3; r++) {
#pragma unroll
            for (int cc = 0; cc < 3; cc++) {
                dG[4 * r + cc] = dAp[4 * r + cc] - dAp[4 * r + 3] * Jj[cc];
                dJ[cc] -= G[4 * r + cc] * dAp[4 * r + 3];
            }
            dG[4 * r + 3] = dAp[4 * r + 3];
        }
    }
#pragma unroll
    for (int k = 0; k < 12; k++) sD[j][k] = dG[k];
    sJ[j][0] = 0.0f; sJ[j][1] = 0.0f; sJ[j][2] = 0.0f;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    float dT[12];
#pragma unroll
    for (int k = 0; k < 12; k++) dT[k] = 0.0f;
    for (int i = J - 1; i >= 0; i--) {                            // reverse chain: children have larger indices
        if (j == i) {
#pragma unroll
            for (int k = 0; k < 12; k++) dG[k] = sD[j][k];
            if (par < 0) {
#pragma unroll
                for (int k = 0; k < 12; k++) dT[k] = dG[k];
            } else {
                float P[12];
#pragma unroll
                for (int k = 0; k < 12; k++) P[k] = sG[par][k];
                // G = P o T:  dT.R = P.R^T dG.R, dT.t = P.R^T dG.t;  dP.R += dG.R T.R^T + dG.t T.t^T, dP.t += dG.t
#pragma unroll
                for (int r = 0; r < 3; r++)
#pragma unroll
                    for (int cc = 0; cc < 4; cc++)
                        dT[4 * r + cc] = (P[r] * dG[cc] + P[4 + r] * dG[4 + cc]) + P[8 + r] * dG[8 + cc];
#pragma unroll
                for (int r = 0; r < 3; r++) {
#pragma unroll
                    for (int cc = 0; cc < 3; cc++)
                        sD[par][4 * r + cc] += ((dG[4 * r] * T[4 * cc] + dG[4 * r + 1] * T[4 * cc + 1]) + dG[4 * r + 2] * T[4 * cc + 2]) +
                                               dG[4 * r + 3] * T[4 * cc + 3];
                    sD[par][4 * r + 3] += dG[4 * r + 3];
                }
                // T.t = J_j - J_parent
#pragma unroll
                for (int k = 0; k < 3; k++) sJ[par][k] -= dT[4 * k + 3];
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    }
    if (!on) return;
    const float M[9] = { dT[0], dT[1], dT[2], dT[4], dT[5], dT[6], dT[8], dT[9], dT[10] };
    float dv[3];
    sg_rodrigues_bwd(v, ro, M, dv);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        dpose[((size_t)b * J + j) * 3 + k] = dv[k];
        if (djoints) djoints[((size_t)b * J + j) * 3 + k] = dJ[k] + dT[4 * k + 3] + sJ[j][k];
    }
}
void sg_launch_joint_transforms(int B, int J, const float *pose, const float *joints, const int *parents, const float *post,
                                const float *dA, float *out0, float *out1, hipStream_t st)
{
    if (B <= 0) return;
    if (dA) hipLaunchKernelGGL(sg_joint_transforms_bwd_kernel, dim3(B), dim3(64), 0, st, J, pose, joints, parents, post, dA, out0, out1);
    else hipLaunchKernelGGL(sg_joint_transforms_fwd_kernel, dim3(B), dim3(64), 0, st, J, pose, joints, parents, post, out0);
}

// ---- launchers ---------------------------------------------------------------------------
void sg_launch_skin_fwd(const SgCam &c, const SgBatch &bt, int P, const SgSkinInputs *in, const float *shs, const float *opacities,
                        const float *scales, SgGeom g, SgBin b, size_t cap, int32_t *radii, float *posed_xyz,
                        float *posed_rotq, float *posed_scales, hipStream_t st)
{
    if (P <= 0) return;
    SgSkin k = { in->J, in->rot_format == SG_ROT_CANON_6D, in->xyz_canon, in->rot_canon, in->lbs_weights, in->A, in->smpl_scale,
                 in->transl, in->ext_trans, in->ext_rot, in->ext_scale };
    const int nblocks = (P + SG_SKIN_THREADS - 1) / SG_SKIN_THREADS;
    dim3 grid(sg_frame_grid(nblocks, bt.K)), block(SG_SKIN_THREADS);
    const int ht = sg_lds_hist(c.gx, c.gy) ? (int)sg_ctr_count((uint32_t)c.gx, (uint32_t)c.gy) : 0;     // histogram words (= tile counters)
#define SG_SF(DD) hipLaunchKernelGGL(sg_skin_fwd_kernel<DD>, grid, block, 0, st, c, bt, P, k, shs, opacities, scales, g, \
                                     b, sg_cap32(cap), radii, posed_xyz, posed_rotq, posed_scales, ht, nblocks)
    sg_prof_begin(SG_K_PREPROCESS_FWD, st);
    switch (c.D) { case 0: SG_SF(0); break; case 1: SG_SF(1); break; case 2: SG_SF(2); break; default: SG_SF(3); break; }
    sg_prof_end(SG_K_PREPROCESS_FWD, st);
#undef SG_SF
}

// dynamic LDS above 64 KB needs the kernel's limit raised (once per kernel and process; J = 64: 88 KB)
static void sg_skin_frames_lds(const void *fn, size_t bytes)
{
    if (bytes > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// per frame: the per-wave slabs, followed (after the slabs of ALL frames) by the SG_RED_GROUPS partial rows of every frame
static size_t sg_skin_slab_rows(int P) { return (size_t)((P + SG_SKIN_THREADS - 1) / SG_SKIN_THREADS) * SG_SKIN_WAVES; }
size_t sg_skin_slab_floats(int P, int K)
{
    const size_t k = (size_t)(K > 0 ? K : 1);
    // (+ the record sums of the K-frame backward: 12 floats per frame and Gaussian, 16-byte aligned behind the partial rows)
    return (sg_skin_slab_rows(P) + SG_RED_GROUPS) * k * (SG_JMAX * 16 + 4) + (k > 1 ? k * 12 * (size_t)P + 4 : 0);
}

void sg_launch_skin_bwd(const SgCam &c, const SgBatch &bt, int P, const SgSkinInputs *in, const float *shs, const float *scales,
                        const int32_t *radii, SgGeom g, SgRec grec, size_t cap, const uint32_t *header, const uint8_t *rec_valid,
                        const float *dposed_xyz_in, const float *dposed_rotq_in, float *slab, float *dL_dxyz_canon, float *dL_drot_canon,
                        float *dL_dscales, float *dL_dopacity, float *dL_dsh, float *dL_dmeans2D, float *dL_dA,
                        float *dL_dtransl, int accumulate, hipStream_t st)
{
    if (P <= 0) return;
    SgSkin k = { in->J, in->rot_format == SG_ROT_CANON_6D, in->xyz_canon, in->rot_canon, in->lbs_weights, in->A, in->smpl_scale,
                 in->transl, nullptr, nullptr, nullptr };
    const int nblocks = (P + SG_SKIN_THREADS - 1) / SG_SKIN_THREADS, stride = SG_JMAX * 16 + 4;
    const size_t slab_frame = sg_skin_slab_rows(P) * stride;
    dim3 grid(nblocks), block(SG_SKIN_THREADS);
#define SG_SB1(DD, AA) hipLaunchKernelGGL((sg_skin_bwd_kernel<DD, AA>), grid, block, 0, st, c, P, k, shs, scales, radii, g,       \
                                     grec, cap, header, rec_valid, dposed_xyz_in, dposed_rotq_in, dL_dxyz_canon,  \
                                     dL_drot_canon, dL_dscales, dL_dopacity, dL_dsh, dL_dmeans2D, slab, stride)
#define SG_SBK(DD, AA) hipLaunchKernelGGL((sg_skin_bwd_frames_kernel<DD, AA>), grid, block, dyn, st, c, bt, P, k, shs, scales, radii, g,       \
                                     grec, cap, header, dposed_xyz_in, dposed_rotq_in, dL_dxyz_canon,  \
                                     dL_drot_canon, dL_dscales, dL_dopacity, dL_dsh, dL_dmeans2D, slab, stride, slab_frame, a9buf)
#define SG_SB2(DD, AA) do { if (bt.K == 1) SG_SB1(DD, AA); else { sg_skin_frames_lds((const void *)sg_skin_bwd_frames_kernel<DD, AA>, dyn); SG_SBK(DD, AA); } } while (0)
#define SG_SB(DD) do { if (accumulate) SG_SB2(DD, true); else SG_SB2(DD, false); } while (0)
    // the frames kernel: sA + 4 waves x (resident weight tile 64 x (J + 1) + scratch 64 x 17) floats of dynamic LDS
    const size_t dyn = ((size_t)SG_JMAX * 16 + (size_t)SG_SKIN_WAVES * (64 * (in->J + 1) + 64 * SG_WSTRIDE)) * sizeof(float);
    sg_prof_begin(SG_K_PREPROCESS_BWD, st);
    // K > 1: the record sums of all frames first, as a kernel of their own (a9 [K][P][12] behind the slabs and partial rows)
    const float4 *a9buf = (const float4 *)(slab + (slab_frame + (size_t)SG_RED_GROUPS * stride) * bt.K);
    if (bt.K > 1)
        sg_launch_record_sums(bt, P, radii, g, grec, cap, header, rec_valid, (float4 *)a9buf, st);
    switch (c.D) { case 0: SG_SB(0); break; case 1: SG_SB(1); break; case 2: SG_SB(2); break; default: SG_SB(3); break; }
    float *part = slab + slab_frame * bt.K;
    hipLaunchKernelGGL(sg_skin_reduce1_kernel, dim3((stride + 63) / 64, SG_RED_GROUPS / 4, bt.K), dim3(256), 0, st, slab,
                       nblocks * SG_SKIN_WAVES, stride, part, slab_frame);
    hipLaunchKernelGGL(sg_skin_reduce2_kernel, dim3((in->J * 16 + 3 + 63) / 64, 1, bt.K), dim3(64), 0, st, part, stride, in->J,
                       dL_dA, dL_dtransl);
    sg_prof_end(SG_K_PREPROCESS_BWD, st);
#undef SG_SB
#undef SG_SB2
#undef SG_SB1
#undef SG_SBK
}

void sg_launch_lbs_fwd(int P, int J, const float *W, const float *A, const float *v, float *T_out, float *verts, hipStream_t st)
{
    if (P <= 0) return;
    hipLaunchKernelGGL(sg_lbs_fwd_kernel, dim3((P + SG_SKIN_THREADS - 1) / SG_SKIN_THREADS), dim3(SG_SKIN_THREADS), 0, st, P, J, W,
                       A, v, T_out, verts);
}
void sg_launch_lbs_bwd(int P, int J, const float *W, const float *A, const float *v, const float *dT, const float *dverts,
                       float *slab, float *dv, float *dA, hipStream_t st)
{
    if (P <= 0) return;
    const int nblocks = (P + SG_SKIN_THREADS - 1) / SG_SKIN_THREADS, stride = SG_JMAX * 16 + 4;
    hipLaunchKernelGGL(sg_lbs_bwd_kernel, dim3(nblocks), dim3(SG_SKIN_THREADS), 0, st, P, J, W, A, v, dT, dverts, dv, slab, stride);
    float *part = slab + (size_t)nblocks * SG_SKIN_WAVES * stride;
    hipLaunchKernelGGL(sg_skin_reduce1_kernel, dim3((stride + 63) / 64, SG_RED_GROUPS / 4), dim3(256), 0, st, slab,
                       nblocks * SG_SKIN_WAVES, stride, part, (size_t)0);
    hipLaunchKernelGGL(sg_skin_reduce2_kernel, dim3((J * 16 + 3 + 63) / 64), dim3(64), 0, st, part, stride, J, dA, (float *)nullptr);
}
