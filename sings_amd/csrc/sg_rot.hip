// Stand-alone rotation conversions, value and gradient, one lane per element.
// Replace the ~10-25 element-wise torch launches each of these is in the reference
// (sings/rec/utils/geometry/rotations.py; call sites sings_hybrid.py:356-357 -- rotation_6d_to_matrix of every
// Gaussian --, :419-428, gs_trainer.py pose parameters) by one launch each way.  Math: sg_rot.h.
#include "sg_rot.h"

template <int NI, int NO, typename F>
__global__ void __launch_bounds__(256) sg_rot_map_kernel(int N, const float *__restrict__ in, float *__restrict__ out, F f)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float x[NI], y[NO];
#pragma unroll
    for (int k = 0; k < NI; k++) x[k] = in[(size_t)i * NI + k];
    f(x, y);
#pragma unroll
    for (int k = 0; k < NO; k++) out[(size_t)i * NO + k] = y[k];
}
// backward: (input x [NI], upstream g [NO]) -> dx [NI]
template <int NI, int NO, typename F>
__global__ void __launch_bounds__(256) sg_rot_grad_kernel(int N, const float *__restrict__ in, const float *__restrict__ gin,
                                                          float *__restrict__ din, F f)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float x[NI], g[NO], d[NI];
#pragma unroll
    for (int k = 0; k < NI; k++) x[k] = in[(size_t)i * NI + k];
#pragma unroll
    for (int k = 0; k < NO; k++) g[k] = gin[(size_t)i * NO + k];
    f(x, g, d);
#pragma unroll
    for (int k = 0; k < NI; k++) din[(size_t)i * NI + k] = d[k];
}

struct SgQ2M { __device__ void operator()(const float *x, float *y) const { sg_q2m(x, y); }
               __device__ void operator()(const float *x, const float *g, float *d) const { sg_q2m_bwd(x, g, d); } };
struct SgR6D { __device__ void operator()(const float *x, float *y) const { sg_r6d2m(x, y); }
               __device__ void operator()(const float *x, const float *g, float *d) const { sg_r6d2m_bwd(x, g, d); } };
struct SgAA2Q { __device__ void operator()(const float *x, float *y) const { sg_aa2q(x, y); }
                __device__ void operator()(const float *x, const float *g, float *d) const { sg_aa2q_bwd(x, g, d); } };
struct SgQ2AA { __device__ void operator()(const float *x, float *y) const { sg_q2aa(x, y); }
                __device__ void operator()(const float *x, const float *g, float *d) const { sg_q2aa_bwd(x, g, d); } };

__global__ void __launch_bounds__(256) sg_qmul_kernel(int N, const float *__restrict__ a, const float *__restrict__ b,
                                                      float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float4 qa = ((const float4 *)a)[i], qb = ((const float4 *)b)[i];
    const float x[4] = { qa.x, qa.y, qa.z, qa.w }, y[4] = { qb.x, qb.y, qb.z, qb.w };
    float o[4];
    sg_qmul(x, y, o);
    ((float4 *)out)[i] = make_float4(o[0], o[1], o[2], o[3]);
}
__global__ void __launch_bounds__(256) sg_qmul_bwd_kernel(int N, const float *__restrict__ a, const float *__restrict__ b,
                                                          const float *__restrict__ g, float *__restrict__ da,
                                                          float *__restrict__ db)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float4 qa = ((const float4 *)a)[i], qb = ((const float4 *)b)[i], gg = ((const float4 *)g)[i];
    const float x[4] = { qa.x, qa.y, qa.z, qa.w }, y[4] = { qb.x, qb.y, qb.z, qb.w }, u[4] = { gg.x, gg.y, gg.z, gg.w };
    float o[4], dx[4], dy[4];
    const float sg = sg_qmul(x, y, o);
    sg_qmul_bwd(x, y, sg, u, dx, dy);
    ((float4 *)da)[i] = make_float4(dx[0], dx[1], dx[2], dx[3]);
    ((float4 *)db)[i] = make_float4(dy[0], dy[1], dy[2], dy[3]);
}

// op: SG_ROT_* of include/sings_hip.h; g == nullptr: value (out [N, NO]); otherwise gradient (out [N, NI])
int sg_launch_rot_map(int op, int N, const float *in, const float *g, float *out, hipStream_t st)
{
    const dim3 grid((N + 255) / 256), block(256);
#define SG_ROT_CASE(OP, NI, NO, FN)                                                                                      \
    case OP:                                                                                                             \
        if (g) hipLaunchKernelGGL((sg_rot_grad_kernel<NI, NO, FN>), grid, block, 0, st, N, in, g, out, FN());            \
        else hipLaunchKernelGGL((sg_rot_map_kernel<NI, NO, FN>), grid, block, 0, st, N, in, out, FN());                  \
        return 0
    switch (op) {
        SG_ROT_CASE(SG_ROT_QUATERNION_TO_MATRIX, 4, 9, SgQ2M);
        SG_ROT_CASE(SG_ROT_6D_TO_MATRIX, 6, 9, SgR6D);
        SG_ROT_CASE(SG_ROT_AXIS_ANGLE_TO_QUATERNION, 3, 4, SgAA2Q);
        SG_ROT_CASE(SG_ROT_QUATERNION_TO_AXIS_ANGLE, 4, 3, SgQ2AA);
    default: return 1;
    }
#undef SG_ROT_CASE
}

void sg_launch_qmul(int N, const float *a, const float *b, const float *g, float *out0, float *out1, hipStream_t st)
{
    const dim3 grid((N + 255) / 256), block(256);
    if (g) hipLaunchKernelGGL(sg_qmul_bwd_kernel, grid, block, 0, st, N, a, b, g, out0, out1);
    else hipLaunchKernelGGL(sg_qmul_kernel, grid, block, 0, st, N, a, b, out0);
}
