// Rotation conversions of SinGS' pose / Gaussian-rotation bookkeeping as device functions:
// value and analytic gradient of the expressions in sings/rec/utils/geometry/rotations.py --
//   quaternion_to_matrix :38-66, quaternion_multiply (raw product :372-390 + standardize :357) :393-407,
//   axis_angle_to_quaternion :482-511, quaternion_to_axis_angle :514-545, rotation_6d_to_matrix :545-566.
// Quaternions are real part first.  Gradients are those torch.autograd gives for the reference expressions
// (torch.where / clamp pass the gradient of the selected branch; F.normalize divides by max(norm, 1e-12)).
// One lane handles one element; used by the stand-alone kernels of sg_rot.hip and by the LBS-fused kernels
// (canonical rotations handed over in 6-D form).
#pragma once
#include "sg_common.h"

// ---- quaternion -> matrix (row-major 3x3); two_s = 2 / |q|^2
__device__ __forceinline__ void sg_q2m(const float q[4], float m[9])
{
    const float r = q[0], i = q[1], j = q[2], k = q[3];
    const float s = 2.0f / (r * r + i * i + j * j + k * k);
    m[0] = 1.0f - s * (j * j + k * k); m[1] = s * (i * j - k * r); m[2] = s * (i * k + j * r);
    m[3] = s * (i * j + k * r); m[4] = 1.0f - s * (i * i + k * k); m[5] = s * (j * k - i * r);
    m[6] = s * (i * k - j * r); m[7] = s * (j * k + i * r); m[8] = 1.0f - s * (i * i + j * j);
}
__device__ __forceinline__ void sg_q2m_bwd(const float q[4], const float g[9], float dq[4])
{
    const float r = q[0], i = q[1], j = q[2], k = q[3];
    const float n = r * r + i * i + j * j + k * k, s = 2.0f / n;
    // dL/ds with the products held fixed, then ds/dq = -s^2 q
    const float ds = -g[0] * (j * j + k * k) + g[1] * (i * j - k * r) + g[2] * (i * k + j * r) + g[3] * (i * j + k * r)
                     - g[4] * (i * i + k * k) + g[5] * (j * k - i * r) + g[6] * (i * k - j * r) + g[7] * (j * k + i * r)
                     - g[8] * (i * i + j * j);
    const float c = -s * s * ds;
    dq[0] = s * (-k * g[1] + j * g[2] + k * g[3] - i * g[5] - j * g[6] + i * g[7]) + c * r;
    dq[1] = s * (-2.0f * i * (g[4] + g[8]) + j * (g[1] + g[3]) + k * (g[2] + g[6]) + r * (g[7] - g[5])) + c * i;
    dq[2] = s * (-2.0f * j * (g[0] + g[8]) + i * (g[1] + g[3]) + k * (g[5] + g[7]) + r * (g[2] - g[6])) + c * j;
    dq[3] = s * (-2.0f * k * (g[0] + g[4]) + i * (g[2] + g[6]) + j * (g[5] + g[7]) + r * (g[3] - g[1])) + c * k;
}

// ---- Hamilton product, then the sign that makes the real part non-negative; returns that sign (+1 / -1)
__device__ __forceinline__ float sg_qmul(const float a[4], const float b[4], float o[4])
{
    const float w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    const float x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    const float y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    const float z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    const float sg = w < 0.0f ? -1.0f : 1.0f;
    o[0] = sg * w; o[1] = sg * x; o[2] = sg * y; o[3] = sg * z;
    return sg;
}
__device__ __forceinline__ void sg_qmul_bwd(const float a[4], const float b[4], float sg, const float gin[4], float da[4],
                                            float db[4])
{
    const float g0 = sg * gin[0], g1 = sg * gin[1], g2 = sg * gin[2], g3 = sg * gin[3];
    da[0] = g0 * b[0] + g1 * b[1] + g2 * b[2] + g3 * b[3];
    da[1] = -g0 * b[1] + g1 * b[0] - g2 * b[3] + g3 * b[2];
    da[2] = -g0 * b[2] + g1 * b[3] + g2 * b[0] - g3 * b[1];
    da[3] = -g0 * b[3] - g1 * b[2] + g2 * b[1] + g3 * b[0];
    db[0] = g0 * a[0] + g1 * a[1] + g2 * a[2] + g3 * a[3];
    db[1] = -g0 * a[1] + g1 * a[0] + g2 * a[3] - g3 * a[2];
    db[2] = -g0 * a[2] - g1 * a[3] + g2 * a[0] + g3 * a[1];
    db[3] = -g0 * a[3] + g1 * a[2] - g2 * a[1] + g3 * a[0];
}

// ---- 6-D -> matrix: rows b1 = a1 / |a1|, b2 = normalised (a2 - (b1.a2) b1), b3 = b1 x b2
#define SG_NORM_EPS 1e-12f
__device__ __forceinline__ void sg_r6d2m(const float d[6], float m[9])
{
    const float n1 = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const float i1 = 1.0f / fmaxf(n1, SG_NORM_EPS);
    const float b1[3] = { d[0] * i1, d[1] * i1, d[2] * i1 };
    const float dt = b1[0] * d[3] + b1[1] * d[4] + b1[2] * d[5];
    const float u[3] = { d[3] - dt * b1[0], d[4] - dt * b1[1], d[5] - dt * b1[2] };
    const float n2 = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    const float i2 = 1.0f / fmaxf(n2, SG_NORM_EPS);
    const float b2[3] = { u[0] * i2, u[1] * i2, u[2] * i2 };
    m[0] = b1[0]; m[1] = b1[1]; m[2] = b1[2];
    m[3] = b2[0]; m[4] = b2[1]; m[5] = b2[2];
    m[6] = b1[1] * b2[2] - b1[2] * b2[1];
    m[7] = b1[2] * b2[0] - b1[0] * b2[2];
    m[8] = b1[0] * b2[1] - b1[1] * b2[0];
}
// y = x / max(|x|, eps): gradient g -> (g - y (y.g)) / |x|, or g / eps where the clamp is active
__device__ __forceinline__ void sg_normalize_bwd(const float y[3], float n, const float g[3], float dx[3])
{
    if (n > SG_NORM_EPS) {
        const float yg = y[0] * g[0] + y[1] * g[1] + y[2] * g[2], inv = 1.0f / n;
        dx[0] = (g[0] - y[0] * yg) * inv; dx[1] = (g[1] - y[1] * yg) * inv; dx[2] = (g[2] - y[2] * yg) * inv;
    } else {
        dx[0] = g[0] / SG_NORM_EPS; dx[1] = g[1] / SG_NORM_EPS; dx[2] = g[2] / SG_NORM_EPS;
    }
}
__device__ __forceinline__ void sg_r6d2m_bwd(const float d[6], const float g[9], float dd[6])
{
    const float n1 = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const float i1 = 1.0f / fmaxf(n1, SG_NORM_EPS);
    const float b1[3] = { d[0] * i1, d[1] * i1, d[2] * i1 };
    const float dt = b1[0] * d[3] + b1[1] * d[4] + b1[2] * d[5];
    const float u[3] = { d[3] - dt * b1[0], d[4] - dt * b1[1], d[5] - dt * b1[2] };
    const float n2 = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    const float i2 = 1.0f / fmaxf(n2, SG_NORM_EPS);
    const float b2[3] = { u[0] * i2, u[1] * i2, u[2] * i2 };
    const float *g1 = g, *g2 = g + 3, *g3 = g + 6;
    // b3 = b1 x b2:  dL/db1 += b2 x g3,  dL/db2 += g3 x b1
    float db1[3] = { g1[0] + (b2[1] * g3[2] - b2[2] * g3[1]), g1[1] + (b2[2] * g3[0] - b2[0] * g3[2]),
                     g1[2] + (b2[0] * g3[1] - b2[1] * g3[0]) };
    const float db2[3] = { g2[0] + (g3[1] * b1[2] - g3[2] * b1[1]), g2[1] + (g3[2] * b1[0] - g3[0] * b1[2]),
                           g2[2] + (g3[0] * b1[1] - g3[1] * b1[0]) };
    float du[3];
    sg_normalize_bwd(b2, n2, db2, du);
    // u = a2 - (b1.a2) b1
    const float b1du = b1[0] * du[0] + b1[1] * du[1] + b1[2] * du[2];
    dd[3] = du[0] - b1du * b1[0]; dd[4] = du[1] - b1du * b1[1]; dd[5] = du[2] - b1du * b1[2];
    db1[0] += -dt * du[0] - b1du * d[3]; db1[1] += -dt * du[1] - b1du * d[4]; db1[2] += -dt * du[2] - b1du * d[5];
    sg_normalize_bwd(b1, n1, db1, dd);
}

// ---- axis-angle -> quaternion: (cos(t/2), a sin(t/2)/t), with the reference's series 1/2 - t^2/48 for |t| < 1e-6
__device__ __forceinline__ void sg_aa2q(const float a[3], float q[4])
{
    const float t = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), h = 0.5f * t;
    const float f = fabsf(t) < 1e-6f ? 0.5f - (t * t) / 48.0f : sinf(h) / t;
    q[0] = cosf(h); q[1] = a[0] * f; q[2] = a[1] * f; q[3] = a[2] * f;
}
__device__ __forceinline__ void sg_aa2q_bwd(const float a[3], const float g[4], float da[3])
{
    const float t = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), h = 0.5f * t;
    const bool small = fabsf(t) < 1e-6f;
    const float f = small ? 0.5f - (t * t) / 48.0f : sinf(h) / t;
    // d f / d t divided by t (the norm's gradient is a / t; 0 at t = 0 as torch defines it)
    const float fp_over_t = small ? -1.0f / 24.0f : (0.5f * cosf(h) - f) / (t * t);
    const float sh_over_t = small ? 0.25f : 0.5f * sinf(h) / t;                  // -(d cos(t/2)/dt) / t
    const float ag = a[0] * g[1] + a[1] * g[2] + a[2] * g[3];
    const float common = t == 0.0f ? 0.0f : ag * fp_over_t - g[0] * sh_over_t;
#pragma unroll
    for (int k = 0; k < 3; k++) da[k] = f * g[1 + k] + a[k] * common;
}

// ---- quaternion -> axis-angle: v / (sin(t/2)/t), t = 2 atan2(|v|, w)
__device__ __forceinline__ void sg_q2aa(const float q[4], float a[3])
{
    const float n = sqrtf(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const float h = atan2f(n, q[0]), t = 2.0f * h;
    const float f = fabsf(t) < 1e-6f ? 0.5f - (t * t) / 48.0f : sinf(h) / t;
    a[0] = q[1] / f; a[1] = q[2] / f; a[2] = q[3] / f;
}
__device__ __forceinline__ void sg_q2aa_bwd(const float q[4], const float g[3], float dq[4])
{
    const float n = sqrtf(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), w = q[0];
    const float h = atan2f(n, w), t = 2.0f * h;
    const bool small = fabsf(t) < 1e-6f;
    const float f = small ? 0.5f - (t * t) / 48.0f : sinf(h) / t;
    const float fp = small ? -t / 24.0f : (0.5f * cosf(h) - f) / t;               // d f / d t
    const float vg = q[1] * g[0] + q[2] * g[1] + q[3] * g[2];
    const float dLdt = -vg * fp / (f * f);                                        // out = v / f(t)
    const float den = n * n + w * w;
    const float dt_dn = den > 0.0f ? 2.0f * w / den : 0.0f, dt_dw = den > 0.0f ? -2.0f * n / den : 0.0f;
    dq[0] = dLdt * dt_dw;
    const float radial = n > 0.0f ? dLdt * dt_dn / n : 0.0f;                      // d|v|/dv = v / |v| (0 at 0)
#pragma unroll
    for (int k = 0; k < 3; k++) dq[1 + k] = g[k] / f + q[1 + k] * radial;
}
