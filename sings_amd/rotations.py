"""Rotation conversions the callers of the render path use: host-side torch, differentiable.

Same names, conventions (quaternions real part first) and branch behaviour as
``sings/rec/utils/geometry/rotations.py`` (pytorch3d-derived): ``quaternion_to_matrix`` :38-66,
``matrix_to_quaternion`` :98-149, ``standardize_quaternion`` :357, ``quaternion_multiply`` :393-407,
``axis_angle_to_quaternion`` :482-511, ``quaternion_to_axis_angle`` :514-545, ``axis_angle_to_matrix`` :450,
``matrix_to_axis_angle`` :466, ``rotation_6d_to_matrix`` :545-566, ``matrix_to_rotation_6d`` :569-585, and the 6-D <->
axis-angle pair the pose optimisation uses (:596-603).  Pinned by tests/golden/rot_cam_golden.npz.
(The device kernels carry their own copies of matrix_to_quaternion / quaternion_multiply, csrc/sg_skin.hip.)
"""
import torch
import torch.nn.functional as F


def quaternion_to_matrix(quaternions):
    q = quaternions
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def _sqrt_positive_part(x):
    ret = torch.zeros_like(x)
    pos = x > 0
    ret[pos] = torch.sqrt(x[pos])
    return ret


class _M2Q(torch.autograd.Function):
    """matrix_to_quaternion on the MI355X: one kernel each way instead of ~25 (sg_matrix_to_quaternion[_backward])."""

    @staticmethod
    def forward(ctx, matrix):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        m = matrix.contiguous()
        N = m.numel() // 9
        q = torch.empty(m.shape[:-2] + (4,), dtype=torch.float32, device=m.device)
        with torch.cuda.device(m.device):
            _lib.check(lib.sg_matrix_to_quaternion(N, C.c_void_p(m.data_ptr()), C.c_void_p(q.data_ptr()),
                                                   C.c_void_p(torch.cuda.current_stream(m.device).cuda_stream)), "matrix_to_quaternion")
        ctx.save_for_backward(m)
        return q

    @staticmethod
    def backward(ctx, dq):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        (m,) = ctx.saved_tensors
        dq = dq.contiguous().float()
        dm = torch.empty_like(m)
        with torch.cuda.device(m.device):
            _lib.check(lib.sg_matrix_to_quaternion_backward(m.numel() // 9, C.c_void_p(m.data_ptr()), C.c_void_p(dq.data_ptr()),
                                                            C.c_void_p(dm.data_ptr()),
                                                            C.c_void_p(torch.cuda.current_stream(m.device).cuda_stream)),
                       "matrix_to_quaternion backward")
        return dm


def matrix_to_quaternion(matrix):
    """rotations.py:98-149.  fp32 matrices on the GPU go through the HIP kernel (per-Gaussian rotations: [N,3,3] with
    N ~ 1e5, sings_hybrid.py:419); anything else (fp64, CPU-side pose bookkeeping of a few joints) through the same
    expression in torch.  On the GPU both give bit-identical fp32 quaternions (within one ulp of the CPU golden G1)."""
    if matrix.is_cuda and matrix.dtype == torch.float32 and matrix.numel() >= 9:
        return _M2Q.apply(matrix)
    return _matrix_to_quaternion_torch(matrix)


def _matrix_to_quaternion_torch(matrix):
    batch_dim = matrix.shape[:-2]
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(matrix.reshape(batch_dim + (9,)), dim=-1)
    q_abs = _sqrt_positive_part(torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22,
                                             1.0 - m00 - m11 + m22], dim=-1))
    quat_by_rijk = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1)], dim=-2)
    flr = torch.tensor(0.1).to(dtype=q_abs.dtype, device=q_abs.device)
    quat_candidates = quat_by_rijk / (2.0 * q_abs[..., None].max(flr))
    return quat_candidates[F.one_hot(q_abs.argmax(dim=-1), num_classes=4) > 0.5, :].reshape(batch_dim + (4,))


def standardize_quaternion(quaternions):
    return torch.where(quaternions[..., 0:1] < 0, -quaternions, quaternions)


def quaternion_raw_multiply(a, b):
    aw, ax, ay, az = torch.unbind(a, -1)
    bw, bx, by, bz = torch.unbind(b, -1)
    return torch.stack((aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw), -1)


def quaternion_multiply(a, b):
    return standardize_quaternion(quaternion_raw_multiply(a, b))


def _sin_half_over_angle(angles, half_angles):
    small = angles.abs() < 1e-6
    out = torch.empty_like(angles)
    out[~small] = torch.sin(half_angles[~small]) / angles[~small]
    out[small] = 0.5 - (angles[small] * angles[small]) / 48          # sin(x/2)/x ~ 1/2 - x^2/48
    return out


def axis_angle_to_quaternion(axis_angle):
    angles = torch.norm(axis_angle, p=2, dim=-1, keepdim=True)
    half = angles * 0.5
    return torch.cat([torch.cos(half), axis_angle * _sin_half_over_angle(angles, half)], dim=-1)


def quaternion_to_axis_angle(quaternions):
    q = quaternions
    norms = torch.norm(q[..., 1:], p=2, dim=-1, keepdim=True)
    half = torch.atan2(norms, q[..., :1])
    angles = 2 * half
    return q[..., 1:] / _sin_half_over_angle(angles, half)


def axis_angle_to_matrix(axis_angle):
    return quaternion_to_matrix(axis_angle_to_quaternion(axis_angle))


def matrix_to_axis_angle(matrix):
    return quaternion_to_axis_angle(matrix_to_quaternion(matrix))


def rotation_6d_to_matrix(d6):
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = F.normalize(a1, dim=-1)
    b2 = F.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1, dim=-1)
    return torch.stack((b1, b2, torch.cross(b1, b2, dim=-1)), dim=-2)


def matrix_to_rotation_6d(matrix):
    return matrix[..., :2, :].clone().reshape(matrix.size()[:-2] + (6,))


def axis_angle_to_rotation_6d(aa):
    return matrix_to_rotation_6d(axis_angle_to_matrix(aa))


def rotation_6d_to_axis_angle(rot6d):
    return matrix_to_axis_angle(rotation_6d_to_matrix(rot6d))
