// Exact-order fp32 device math for the per-Gaussian kernels.
//
// Everything here is written one IEEE operation at a time and the library is compiled with
// -ffp-contract=off and correctly rounded fp32 divide/sqrt, so that integer outcomes (radius,
// tile rectangle, depth key bits) are reproducible bit for bit against the CPU oracle.
// Algorithm: forward/backward preprocess of the rasterizer SinGS calls at
// sings/rec/renderer/gs_renderer_single.py:87-95 (digest: SURVEY.md App. A.1 / A.5).
#pragma once
#include "sg_common.h"

#define SG_C0 0.28209479177387814
#define SG_C1 0.4886025119029199
#define SG_C2_0 1.0925484305920792
#define SG_C2_1 -1.0925484305920792
#define SG_C2_2 0.31539156525252005
#define SG_C2_3 -1.0925484305920792
#define SG_C2_4 0.5462742152960396
#define SG_C3_0 -0.5900435899266435
#define SG_C3_1 2.890611442640554
#define SG_C3_2 -0.4570457994644658
#define SG_C3_3 0.3731763325901154
#define SG_C3_4 -0.4570457994644658
#define SG_C3_5 1.445305721320277
#define SG_C3_6 -0.5900435899266435

__device__ __forceinline__ void sg_xf4x3(const float p[3], const float *__restrict__ m, float o[3])
{
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
__device__ __forceinline__ void sg_xf4x4(const float p[3], const float *__restrict__ m, float o[4])
{
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

// rotation matrix (row-major) of the UN-normalised quaternion (r,x,y,z)
__device__ __forceinline__ void sg_quat_to_R(const float q[4], float R[9])
{
    float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1.0f - 2.0f * (y * y + z * z);
    R[1] = 2.0f * (x * y - r * z);
    R[2] = 2.0f * (x * z + r * y);
    R[3] = 2.0f * (x * y + r * z);
    R[4] = 1.0f - 2.0f * (x * x + z * z);
    R[5] = 2.0f * (y * z - r * x);
    R[6] = 2.0f * (x * z - r * y);
    R[7] = 2.0f * (y * z + r * x);
    R[8] = 1.0f - 2.0f * (x * x + y * y);
}

// Sigma = R S^2 R^T as M^T M, M[k][i] = s_k R[i][k]
__device__ __forceinline__ void sg_cov3d(const float scale[3], float mod, const float q[4], float c6[6])
{
    float R[9], M[9];
    sg_quat_to_R(q, R);
    float s[3] = { mod * scale[0], mod * scale[1], mod * scale[2] };
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int i = 0; i < 3; i++) M[3 * k + i] = s[k] * R[3 * i + k];
    float S[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            S[3 * i + j] = M[0 + i] * M[0 + j] + M[3 + i] * M[3 + j] + M[6 + i] * M[6 + j];
    c6[0] = S[0]; c6[1] = S[1]; c6[2] = S[2]; c6[3] = S[4]; c6[4] = S[5]; c6[5] = S[8];
}

// Mm = J * Rv (2x3); tc = fov-clamped view-space point
__device__ __forceinline__ void sg_proj_jac(const float t[3], float fx, float fy, float tanfovx,
                                            float tanfovy, const float *__restrict__ view,
                                            float Mm[6], float tc[3], bool &xin, bool &yin)
{
    float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
    float txtz = t[0] / t[2], tytz = t[1] / t[2];
    xin = !(txtz < -limx || txtz > limx);
    yin = !(tytz < -limy || tytz > limy);
    float cx = txtz < -limx ? -limx : txtz; cx = cx > limx ? limx : cx;
    float cy = tytz < -limy ? -limy : tytz; cy = cy > limy ? limy : cy;
    tc[0] = cx * t[2]; tc[1] = cy * t[2]; tc[2] = t[2];
    float j00 = fx / tc[2], j02 = -(fx * tc[0]) / (tc[2] * tc[2]);
    float j11 = fy / tc[2], j12 = -(fy * tc[1]) / (tc[2] * tc[2]);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        Mm[k] = j00 * view[0 + 4 * k] + j02 * view[2 + 4 * k];
        Mm[3 + k] = j11 * view[1 + 4 * k] + j12 * view[2 + 4 * k];
    }
}

__device__ __forceinline__ void sg_cov2d(const float Mm[6], const float c6[6], float abc[3])
{
    float V[9] = { c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5] };
    float tmp[6];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            tmp[3 * a + j] = Mm[3 * a + 0] * V[0 + j] + Mm[3 * a + 1] * V[3 + j] + Mm[3 * a + 2] * V[6 + j];
    abc[0] = tmp[0] * Mm[0] + tmp[1] * Mm[1] + tmp[2] * Mm[2];
    abc[1] = tmp[0] * Mm[3] + tmp[1] * Mm[4] + tmp[2] * Mm[5];
    abc[2] = tmp[3] * Mm[3] + tmp[4] * Mm[4] + tmp[5] * Mm[5];
    abc[0] += 0.3f;
    abc[2] += 0.3f;
}

// SH basis (polynomial of sings/rec/utils/visualize/spherical_harmonics.py:87-113)
template <int D> __device__ __forceinline__ void sg_sh_basis(const float d[3], float b[16])
{
    float x = d[0], y = d[1], z = d[2];
    b[0] = (float)SG_C0;
    if (D > 0) {
        b[1] = -(float)SG_C1 * y; b[2] = (float)SG_C1 * z; b[3] = -(float)SG_C1 * x;
        if (D > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = (float)SG_C2_0 * xy;
            b[5] = (float)SG_C2_1 * yz;
            b[6] = (float)SG_C2_2 * (2.0f * zz - xx - yy);
            b[7] = (float)SG_C2_3 * xz;
            b[8] = (float)SG_C2_4 * (xx - yy);
            if (D > 2) {
                b[9] = (float)SG_C3_0 * y * (3.0f * xx - yy);
                b[10] = (float)SG_C3_1 * xy * z;
                b[11] = (float)SG_C3_2 * y * (4.0f * zz - xx - yy);
                b[12] = (float)SG_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                b[13] = (float)SG_C3_4 * x * (4.0f * zz - xx - yy);
                b[14] = (float)SG_C3_5 * z * (xx - yy);
                b[15] = (float)SG_C3_6 * x * (xx - 3.0f * yy);
            }
        }
    }
}

// d(basis_k)/d(dir): db[3*k + axis]
template <int D> __device__ __forceinline__ void sg_sh_basis_grad(const float d[3], float db[48])
{
    float x = d[0], y = d[1], z = d[2];
#pragma unroll
    for (int i = 0; i < 48; i++) db[i] = 0.0f;
    if (D > 0) {
        db[3 * 1 + 1] = -(float)SG_C1; db[3 * 2 + 2] = (float)SG_C1; db[3 * 3 + 0] = -(float)SG_C1;
        if (D > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y;
            db[3 * 4 + 0] = (float)SG_C2_0 * y; db[3 * 4 + 1] = (float)SG_C2_0 * x;
            db[3 * 5 + 1] = (float)SG_C2_1 * z; db[3 * 5 + 2] = (float)SG_C2_1 * y;
            db[3 * 6 + 0] = (float)SG_C2_2 * -2.0f * x; db[3 * 6 + 1] = (float)SG_C2_2 * -2.0f * y;
            db[3 * 6 + 2] = (float)SG_C2_2 * 4.0f * z;
            db[3 * 7 + 0] = (float)SG_C2_3 * z; db[3 * 7 + 2] = (float)SG_C2_3 * x;
            db[3 * 8 + 0] = (float)SG_C2_4 * 2.0f * x; db[3 * 8 + 1] = (float)SG_C2_4 * -2.0f * y;
            if (D > 2) {
                db[3 * 9 + 0] = (float)SG_C3_0 * 6.0f * xy;
                db[3 * 9 + 1] = (float)SG_C3_0 * (3.0f * xx - 3.0f * yy);
                db[3 * 10 + 0] = (float)SG_C3_1 * y * z; db[3 * 10 + 1] = (float)SG_C3_1 * x * z;
                db[3 * 10 + 2] = (float)SG_C3_1 * xy;
                db[3 * 11 + 0] = (float)SG_C3_2 * -2.0f * xy;
                db[3 * 11 + 1] = (float)SG_C3_2 * (4.0f * zz - xx - 3.0f * yy);
                db[3 * 11 + 2] = (float)SG_C3_2 * 8.0f * y * z;
                db[3 * 12 + 0] = (float)SG_C3_3 * -6.0f * x * z;
                db[3 * 12 + 1] = (float)SG_C3_3 * -6.0f * y * z;
                db[3 * 12 + 2] = (float)SG_C3_3 * (6.0f * zz - 3.0f * xx - 3.0f * yy);
                db[3 * 13 + 0] = (float)SG_C3_4 * (4.0f * zz - 3.0f * xx - yy);
                db[3 * 13 + 1] = (float)SG_C3_4 * -2.0f * xy;
                db[3 * 13 + 2] = (float)SG_C3_4 * 8.0f * x * z;
                db[3 * 14 + 0] = (float)SG_C3_5 * 2.0f * x * z;
                db[3 * 14 + 1] = (float)SG_C3_5 * -2.0f * y * z;
                db[3 * 14 + 2] = (float)SG_C3_5 * (xx - yy);
                db[3 * 15 + 0] = (float)SG_C3_6 * (3.0f * xx - 3.0f * yy);
                db[3 * 15 + 1] = (float)SG_C3_6 * -6.0f * xy;
            }
        }
    }
}

__device__ __forceinline__ int sg_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// tile rectangle of a splat (getRect of the upstream rasterizer, C truncation then clamp)
__device__ __forceinline__ void sg_rect(float px, float py, int mr, int gx, int gy, int &x0, int &y0,
                                        int &x1, int &y1)
{
    x0 = sg_clampi((int)((px - (float)mr) / 16.0f), 0, gx);
    y0 = sg_clampi((int)((py - (float)mr) / 16.0f), 0, gy);
    x1 = sg_clampi((int)((px + (float)mr + 15.0f) / 16.0f), 0, gx);
    y1 = sg_clampi((int)((py + (float)mr + 15.0f) / 16.0f), 0, gy);
}
