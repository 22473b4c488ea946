// Forward / backward per-Gaussian ("preprocess") kernels.
//
// Replaces preprocessCUDA<3> (+ computeCov2DCUDA in backward) of the rasterizer the reference
// calls at sings/rec/renderer/gs_renderer_single.py:87-95; algorithm per SURVEY.md App. A.1/A.5.
// One lane per Gaussian; 256-thread workgroups; grid = ceil(P/256) >> 256 CUs at P >= 1e5.
// HBM-bound streaming kernel: reads `in` bytes per Gaussian, writes 56 B of projected record.
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

template <int D>
__global__ void __launch_bounds__(256)
sg_preprocess_fwd_kernel(SgCam c, int P, const float *__restrict__ means3D,
                         const float *__restrict__ shs, const float *__restrict__ colors_precomp,
                         const float *__restrict__ opacities, const float *__restrict__ scales,
                         const float *__restrict__ rotations, const float *__restrict__ cov3D_precomp,
                         SgGeom g, uint32_t *__restrict__ header, int32_t *__restrict__ radii)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool live = idx < P;
    int mr = 0;
    uint32_t tt = 0, clampbits = 0;
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    float pix[2] = { 0, 0 }, conic[3] = { 0, 0, 0 }, rgb[3] = { 0, 0, 0 }, depth = 0, opac = 0;
    if (live) {
        float p[3] = { means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2] };
        float pv[3];
        sg_xf4x3(p, c.view, pv);
        bool ok = pv[2] > 0.2f;
        if (ok) {
            float ph[4];
            sg_xf4x4(p, c.proj, ph);
            float pw = 1.0f / (ph[3] + 0.0000001f);
            float ppx = ph[0] * pw, ppy = ph[1] * pw;
            float c6[6];
            if (cov3D_precomp) {
#pragma unroll
                for (int k = 0; k < 6; k++) c6[k] = cov3D_precomp[6 * idx + k];
            } else {
                float s3[3] = { scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2] };
                float q[4] = { rotations[4 * idx], rotations[4 * idx + 1], rotations[4 * idx + 2], rotations[4 * idx + 3] };
                sg_cov3d(s3, c.mod, q, c6);
            }
            float Mm[6], tc[3], abc[3]; bool xin, yin;
            sg_proj_jac(pv, c.fx, c.fy, c.tanfovx, c.tanfovy, c.view, Mm, tc, xin, yin);
            sg_cov2d(Mm, c6, abc);
            float det = abc[0] * abc[2] - abc[1] * abc[1];
            ok = det != 0.0f;
            if (ok) {
                float det_inv = 1.0f / det;
                conic[0] = abc[2] * det_inv; conic[1] = -abc[1] * det_inv; conic[2] = abc[0] * det_inv;
                float mid = 0.5f * (abc[0] + abc[2]);
                float dd = mid * mid - det; dd = dd < 0.1f ? 0.1f : dd;
                float sq = sqrtf(dd);
                float l1 = mid + sq, l2 = mid - sq;
                float lm = l1 > l2 ? l1 : l2;
                float my_radius = ceilf(3.0f * sqrtf(lm));
                pix[0] = ((ppx + 1.0f) * (float)c.W - 1.0f) * 0.5f;
                pix[1] = ((ppy + 1.0f) * (float)c.H - 1.0f) * 0.5f;
                int r = (int)my_radius;
                sg_rect(pix[0], pix[1], r, c.gx, c.gy, x0, y0, x1, y1);
                uint32_t area = (uint32_t)((x1 - x0) * (y1 - y0));
                if (area != 0) {
                    mr = r; tt = area; depth = pv[2]; opac = opacities[idx];
                    if (colors_precomp) {
                        rgb[0] = colors_precomp[3 * idx]; rgb[1] = colors_precomp[3 * idx + 1]; rgb[2] = colors_precomp[3 * idx + 2];
                    } else {
                        float dir[3] = { p[0] - c.campos[0], p[1] - c.campos[1], p[2] - c.campos[2] };
                        float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
                        dir[0] = dir[0] / len; dir[1] = dir[1] / len; dir[2] = dir[2] / len;
                        sg_eval_sh<D>(shs + (size_t)idx * c.M * 3, dir, rgb, clampbits);
                    }
                }
            }
        }
    }
    // wave-aggregated slot allocation for the backward gradient records (Gaussian-major)
    uint32_t incl = tt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    uint32_t total = __shfl(incl, 63, 64);
    uint32_t base = 0;
    if (lane == 63 && total) base = atomicAdd(&header[2], total);
    base = __shfl(base, 63, 64);
    if (live) {
        uint32_t goff = base + incl - tt;
        radii[idx] = mr;
        g.recA[idx] = make_float4(pix[0], pix[1], conic[0], conic[1]);
        g.recB[idx] = make_float4(conic[2], opac, rgb[0], rgb[1]);
        g.recC[idx] = make_float4(rgb[2], __uint_as_float(goff),
                                  __uint_as_float((uint32_t)x0 | ((uint32_t)y0 << 16)),
                                  __uint_as_float((uint32_t)(x1 - x0) | ((uint32_t)(y1 - y0) << 16)));
        g.depth[idx] = depth;
        g.flags[idx] = clampbits;
    }
}

void sg_launch_preprocess_fwd(const SgCam &c, int P, const float *means3D, const float *shs,
                              const float *colors_precomp, const float *opacities, const float *scales,
                              const float *rotations, const float *cov3D_precomp, SgGeom g, SgBin b,
                              int32_t *radii, hipStream_t st)
{
    if (P <= 0) return;
    dim3 grid((P + 255) / 256), block(256);
#define SG_PP(DD) hipLaunchKernelGGL(sg_preprocess_fwd_kernel<DD>, grid, block, 0, st, c, P, means3D, shs, \
                                     colors_precomp, opacities, scales, rotations, cov3D_precomp, g, b.header, radii)
    int D = colors_precomp ? 0 : c.D;
    sg_prof_begin(SG_K_PREPROCESS_FWD, st);
    switch (D) { case 0: SG_PP(0); break; case 1: SG_PP(1); break; case 2: SG_PP(2); break; default: SG_PP(3); break; }
    sg_prof_end(SG_K_PREPROCESS_FWD, st);
#undef SG_PP
}

// ------------------------------------------------------------------------------------------
// Backward: sums the per-(tile,Gaussian) gradient records of this Gaussian in a fixed order
// (deterministic, no atomics), then chains through conic -> cov2D -> cov3D / mean, the
// projection, the SH colour and scale/rotation.
template <int D>
__global__ void __launch_bounds__(256)
sg_preprocess_bwd_kernel(SgCam c, int P, const float *__restrict__ means3D, const float *__restrict__ shs,
                         const float *__restrict__ colors_precomp, const float *__restrict__ scales,
                         const float *__restrict__ rotations, const float *__restrict__ cov3D_precomp,
                         const int32_t *__restrict__ radii, SgGeom g, const float4 *__restrict__ grec,
                         size_t cap, float *__restrict__ dL_dmeans3D, float *__restrict__ dL_dmeans2D,
                         float *__restrict__ dL_dsh, float *__restrict__ dL_dcolors,
                         float *__restrict__ dL_dopacity, float *__restrict__ dL_dscales,
                         float *__restrict__ dL_drots, float *__restrict__ dL_dcov3D)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    float dmean[3] = { 0, 0, 0 }, g2[2] = { 0, 0 }, dop = 0, dcol[3] = { 0, 0, 0 };
    float dsc[3] = { 0, 0, 0 }, drot[4] = { 0, 0, 0, 0 }, g6[6] = { 0, 0, 0, 0, 0, 0 };
    const bool vis = radii[idx] > 0;
    const int Mrows = c.M;
    float p[3];
    uint32_t clampbits = 0;
    if (vis) {
        float4 rc = g.recC[idx];
        uint32_t goff = __float_as_uint(rc.y), wh = __float_as_uint(rc.w);
        uint32_t tt = (wh & 0xffffu) * (wh >> 16);
        float a9[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
        for (uint32_t k = 0; k < tt; k++) {
            size_t r = (size_t)goff + k;
            if (r >= cap) break;
            float4 r0 = grec[3 * r], r1 = grec[3 * r + 1], r2 = grec[3 * r + 2];
            a9[0] += r0.x; a9[1] += r0.y; a9[2] += r0.z; a9[3] += r0.w;
            a9[4] += r1.x; a9[5] += r1.y; a9[6] += r1.z; a9[7] += r1.w; a9[8] += r2.x;
        }
        // record layout: mean2D.x, mean2D.y, conic.x, conic.y, conic.w, opacity, color rgb
        g2[0] = a9[0]; g2[1] = a9[1];
        float dLx = a9[2], dLy = a9[3], dLz = a9[4];
        dop = a9[5]; dcol[0] = a9[6]; dcol[1] = a9[7]; dcol[2] = a9[8];
        clampbits = g.flags[idx];
        p[0] = means3D[3 * idx]; p[1] = means3D[3 * idx + 1]; p[2] = means3D[3 * idx + 2];
        float c6[6], s3[3] = { 0, 0, 0 }, q[4] = { 0, 0, 0, 0 };
        if (cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; k++) c6[k] = cov3D_precomp[6 * idx + k];
        } else {
            s3[0] = scales[3 * idx]; s3[1] = scales[3 * idx + 1]; s3[2] = scales[3 * idx + 2];
            q[0] = rotations[4 * idx]; q[1] = rotations[4 * idx + 1]; q[2] = rotations[4 * idx + 2]; q[3] = rotations[4 * idx + 3];
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
        if (!cov3D_precomp) {
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
                dsc[k] = c.mod * 2.0f * s[k] * rgr;
#pragma unroll
                for (int a2 = 0; a2 < 3; a2++) dR[3 * a2 + k] = 2.0f * GR[3 * a2 + k] * s[k] * s[k];
            }
            float r = q[0], x = q[1], y = q[2], z = q[3];
            drot[0] = 2.0f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
            drot[1] = 2.0f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.0f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2.0f * x * dR[8]);
            drot[2] = 2.0f * (-2.0f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2.0f * y * dR[8]);
            drot[3] = 2.0f * (-2.0f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.0f * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
        }
    }
    // ---- SH backward (+ its contribution to dL/dmean) and dL/dsh rows (all M rows written)
    if (dL_dsh) {
        float *o = dL_dsh + (size_t)idx * Mrows * 3;
        constexpr int nc = (D + 1) * (D + 1);
        if (vis) {
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
            const float *sh = shs + (size_t)idx * Mrows * 3;
            float ddir[3] = { 0, 0, 0 };
#pragma unroll
            for (int k = 0; k < nc; k++)
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    o[3 * k + ch] = bas[k] * dRGB[ch];
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
            for (int k = nc * 3; k < Mrows * 3; k++) o[k] = 0.0f;
        } else {
            for (int k = 0; k < Mrows * 3; k++) o[k] = 0.0f;
        }
    }
    dL_dmeans3D[3 * idx] = dmean[0]; dL_dmeans3D[3 * idx + 1] = dmean[1]; dL_dmeans3D[3 * idx + 2] = dmean[2];
    dL_dmeans2D[3 * idx] = g2[0]; dL_dmeans2D[3 * idx + 1] = g2[1]; dL_dmeans2D[3 * idx + 2] = 0.0f;
    dL_dopacity[idx] = dop;
    if (dL_dcolors) { dL_dcolors[3 * idx] = dcol[0]; dL_dcolors[3 * idx + 1] = dcol[1]; dL_dcolors[3 * idx + 2] = dcol[2]; }
    if (dL_dscales) { dL_dscales[3 * idx] = dsc[0]; dL_dscales[3 * idx + 1] = dsc[1]; dL_dscales[3 * idx + 2] = dsc[2]; }
    if (dL_drots) { dL_drots[4 * idx] = drot[0]; dL_drots[4 * idx + 1] = drot[1]; dL_drots[4 * idx + 2] = drot[2]; dL_drots[4 * idx + 3] = drot[3]; }
    if (dL_dcov3D) {
#pragma unroll
        for (int k = 0; k < 6; k++) dL_dcov3D[6 * idx + k] = g6[k];
    }
}

void sg_launch_preprocess_bwd(const SgCam &c, int P, const float *means3D, const float *shs,
                              const float *colors_precomp, const float *opacities, const float *scales,
                              const float *rotations, const float *cov3D_precomp,
                              const int32_t *radii, SgGeom g, const float *grec, size_t cap,
                              float *dL_dmeans3D, float *dL_dmeans2D, float *dL_dsh,
                              float *dL_dcolors, float *dL_dopacity, float *dL_dscales,
                              float *dL_drots, float *dL_dcov3D, hipStream_t st)
{
    (void)opacities;
    if (P <= 0) return;
    dim3 grid((P + 255) / 256), block(256);
#define SG_PB(DD) hipLaunchKernelGGL(sg_preprocess_bwd_kernel<DD>, grid, block, 0, st, c, P, means3D, shs, \
                                     colors_precomp, scales, rotations, cov3D_precomp, radii, g,            \
                                     (const float4 *)grec, cap, dL_dmeans3D, dL_dmeans2D, dL_dsh, dL_dcolors, \
                                     dL_dopacity, dL_dscales, dL_drots, dL_dcov3D)
    int D = shs ? c.D : 0;
    sg_prof_begin(SG_K_PREPROCESS_BWD, st);
    switch (D) { case 0: SG_PB(0); break; case 1: SG_PB(1); break; case 2: SG_PB(2); break; default: SG_PB(3); break; }
    sg_prof_end(SG_K_PREPROCESS_BWD, st);
#undef SG_PB
}
