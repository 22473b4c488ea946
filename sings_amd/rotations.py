"""Rotation conversions on the MI355X: the functions of
``sings/rec/utils/geometry/rotations.py`` the render path and the pose optimisation call, same names and conventions
(quaternions real part first), each ONE kernel launch forward and one backward (csrc/sg_rot.h / sg_rot.hip,
``sg_rotation_convert[_backward]``, ``sg_quaternion_multiply[_backward]``, ``sg_matrix_to_quaternion[_backward]``):

    quaternion_to_matrix :38-66        matrix_to_quaternion :98-149       standardize_quaternion :357
    quaternion_multiply :393-407       axis_angle_to_quaternion :482-511  quaternion_to_axis_angle :514-545
    axis_angle_to_matrix :450          matrix_to_axis_angle :466          rotation_6d_to_matrix :545-566
    matrix_to_rotation_6d :569-585     axis_angle_to_rotation_6d / rotation_6d_to_axis_angle :596-603

Inputs are fp32 tensors on the GPU with any leading batch shape; gradients are what torch.autograd gives for the
reference expressions.  There is NO CPU / eager path: host tensors raise (the torch restatement used to check these
kernels lives in oracle/rotations_oracle.py and is test infrastructure).
"""
import ctypes as C

import torch

from . import _lib

_Q2M, _R6D2M, _AA2Q, _Q2AA = 0, 1, 2, 3           # SG_ROT_* of include/sings_hip.h
_WIDTH = {_Q2M: (4, 9), _R6D2M: (6, 9), _AA2Q: (3, 4), _Q2AA: (4, 3)}


def _check(t, width, name):
    if not torch.is_tensor(t):
        raise TypeError(f"{name} must be a tensor")
    if not t.is_cuda:
        raise RuntimeError(f"sings_amd.rotations.{name}: tensors must live on the MI355X (no CPU fallback; got {t.device})")
    if t.dtype != torch.float32:
        raise RuntimeError(f"sings_amd.rotations.{name}: float32 only (got {t.dtype})")
    if t.shape[-1] != width:
        raise ValueError(f"sings_amd.rotations.{name}: last dimension must be {width}, got {tuple(t.shape)}")
    return t.contiguous()


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr())


class _Convert(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, op):
        lib = _lib.load()
        ni, no = _WIDTH[op]
        n = x.numel() // ni
        out = torch.empty(x.shape[:-1] + (no,), dtype=torch.float32, device=x.device)
        if n:
            with torch.cuda.device(x.device):
                _lib.check(lib.sg_rotation_convert(op, n, _p(x), _p(out), _stream(x.device)), "rotation convert")
        ctx.op = op
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        ni, _ = _WIDTH[ctx.op]
        n = x.numel() // ni
        g = g.contiguous().float()
        dx = torch.empty_like(x)
        if n:
            with torch.cuda.device(x.device):
                _lib.check(lib.sg_rotation_convert_backward(ctx.op, n, _p(x), _p(g), _p(dx), _stream(x.device)),
                           "rotation convert backward")
        return dx, None


class _M2Q(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m):
        lib = _lib.load()
        n = m.numel() // 9
        q = torch.empty(m.shape[:-2] + (4,), dtype=torch.float32, device=m.device)
        if n:
            with torch.cuda.device(m.device):
                _lib.check(lib.sg_matrix_to_quaternion(n, _p(m), _p(q), _stream(m.device)), "matrix_to_quaternion")
        ctx.save_for_backward(m)
        return q

    @staticmethod
    def backward(ctx, dq):
        lib = _lib.load()
        (m,) = ctx.saved_tensors
        dq = dq.contiguous().float()
        dm = torch.empty_like(m)
        if m.numel():
            with torch.cuda.device(m.device):
                _lib.check(lib.sg_matrix_to_quaternion_backward(m.numel() // 9, _p(m), _p(dq), _p(dm), _stream(m.device)),
                           "matrix_to_quaternion backward")
        return dm


class _QMul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        lib = _lib.load()
        out = torch.empty_like(a)
        if a.numel():
            with torch.cuda.device(a.device):
                _lib.check(lib.sg_quaternion_multiply(a.numel() // 4, _p(a), _p(b), _p(out), _stream(a.device)),
                           "quaternion_multiply")
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        a, b = ctx.saved_tensors
        g = g.contiguous().float()
        da, db = torch.empty_like(a), torch.empty_like(b)
        if a.numel():
            with torch.cuda.device(a.device):
                _lib.check(lib.sg_quaternion_multiply_backward(a.numel() // 4, _p(a), _p(b), _p(g), _p(da), _p(db),
                                                               _stream(a.device)), "quaternion_multiply backward")
        return da, db


def quaternion_to_matrix(quaternions):
    q = _check(quaternions, 4, "quaternion_to_matrix")
    return _Convert.apply(q, _Q2M).reshape(q.shape[:-1] + (3, 3))


def matrix_to_quaternion(matrix):
    if matrix.shape[-2:] != (3, 3):
        raise ValueError(f"Invalid rotation matrix shape {tuple(matrix.shape)}.")
    m = _check(matrix.reshape(matrix.shape[:-2] + (9,)), 9, "matrix_to_quaternion")
    return _M2Q.apply(m.reshape(matrix.shape))


def quaternion_multiply(a, b):
    """standardize(a * b), broadcasting the leading dimensions like the reference's torch expression."""
    a = _check(a, 4, "quaternion_multiply"); b = _check(b, 4, "quaternion_multiply")
    if a.shape != b.shape:
        a, b = torch.broadcast_tensors(a, b)
        a, b = a.contiguous(), b.contiguous()
    return _QMul.apply(a, b)


def standardize_quaternion(quaternions):
    q = _check(quaternions, 4, "standardize_quaternion")
    return torch.where(q[..., 0:1] < 0, -q, q)


def rotation_6d_to_matrix(d6):
    d = _check(d6, 6, "rotation_6d_to_matrix")
    return _Convert.apply(d, _R6D2M).reshape(d.shape[:-1] + (3, 3))


def matrix_to_rotation_6d(matrix):
    if not matrix.is_cuda:
        raise RuntimeError("sings_amd.rotations.matrix_to_rotation_6d: tensors must live on the MI355X (no CPU fallback)")
    return matrix[..., :2, :].clone().reshape(matrix.size()[:-2] + (6,))      # a copy of the first two rows: no arithmetic


def axis_angle_to_quaternion(axis_angle):
    return _Convert.apply(_check(axis_angle, 3, "axis_angle_to_quaternion"), _AA2Q)


def quaternion_to_axis_angle(quaternions):
    return _Convert.apply(_check(quaternions, 4, "quaternion_to_axis_angle"), _Q2AA)


def axis_angle_to_matrix(axis_angle):
    return quaternion_to_matrix(axis_angle_to_quaternion(axis_angle))


def matrix_to_axis_angle(matrix):
    return quaternion_to_axis_angle(matrix_to_quaternion(matrix))


def axis_angle_to_rotation_6d(aa):
    return matrix_to_rotation_6d(axis_angle_to_matrix(aa))


def rotation_6d_to_axis_angle(rot6d):
    return matrix_to_axis_angle(rotation_6d_to_matrix(rot6d))
