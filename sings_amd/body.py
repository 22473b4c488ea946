"""Joint transforms of an SMPL-shaped kinematic tree, in torch (differentiable, runs on the MI355X).

Host-side producer of the ``A`` the fused LBS kernels consume.  Mirrors the math of the
reference's vendored SMPL chain -- ``batch_rodrigues`` (sings/rec/utils/body_model/smpl.py:415-446) and
``batch_rigid_transform`` (:462-513) -- WITHOUT the template work the reference throws away every step (blend
shapes, the J_regressor einsum over 110k vertices and the full template skinning): the rest
joints are computed once per shape and cached by the caller.  Gradients w.r.t. the pose flow through autograd, so
``dL/dA`` from ``sg_skinned_backward`` reaches ``body_pose`` / ``global_orient``.
"""
import torch

SMPL_PARENTS = (-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21)   # smpl_layer.py:269-272


def rodrigues(rot_vecs):
    """[J,3] axis-angle -> [J,3,3]; angle = |theta + 1e-8| as the reference does."""
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    d = rot_vecs / angle
    c, s = torch.cos(angle)[:, :, None], torch.sin(angle)[:, :, None]
    z = torch.zeros_like(d[:, 0:1])
    K = torch.cat([z, -d[:, 2:3], d[:, 1:2], d[:, 2:3], z, -d[:, 0:1], -d[:, 1:2], d[:, 0:1], z], 1).view(-1, 3, 3)
    eye = torch.eye(3, dtype=rot_vecs.dtype, device=rot_vecs.device)[None]
    return eye + s * K + (1 - c) * torch.bmm(K, K)


class _JointTransforms(torch.autograd.Function):
    """The whole chain in ONE launch each way (sg_joint_transforms / _backward): one wave per frame, lane = joint."""

    @staticmethod
    def forward(ctx, pose, joints_rest, parents_t, post):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        B, J = int(pose.shape[0]), int(joints_rest.shape[0])
        pose = pose.contiguous().float(); jr = joints_rest.contiguous().float()
        post = None if post is None else post.contiguous().float()
        A = torch.empty((B, J, 4, 4), dtype=torch.float32, device=pose.device)
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        with torch.cuda.device(pose.device):
            _lib.check(lib.sg_joint_transforms(B, J, p(pose), p(jr), p(parents_t), p(post), p(A),
                                               C.c_void_p(torch.cuda.current_stream(pose.device).cuda_stream)), "joint transforms")
        ctx.save_for_backward(pose, jr, parents_t, post if post is not None else torch.empty(0, device=pose.device))
        ctx.has_post = post is not None
        return A

    @staticmethod
    def backward(ctx, dA):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        pose, jr, parents_t, post = ctx.saved_tensors
        B, J = int(pose.shape[0]), int(jr.shape[0])
        dA = dA.contiguous().float()
        dpose = torch.empty_like(pose)
        dj = torch.empty((B, J, 3), dtype=torch.float32, device=pose.device) if ctx.needs_input_grad[1] else None
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        with torch.cuda.device(pose.device):
            _lib.check(lib.sg_joint_transforms_backward(B, J, p(pose), p(jr), p(parents_t), p(post) if ctx.has_post else None, p(dA),
                                                        p(dpose), p(dj), C.c_void_p(torch.cuda.current_stream(pose.device).cuda_stream)),
                       "joint transforms backward")
        return dpose, None if dj is None else dj.sum(0), None, None


_parents_cache = {}


def _parents_tensor(parents, device):
    key = (tuple(int(x) for x in parents), str(device))
    t = _parents_cache.get(key)
    if t is None:
        if key[0][0] >= 0 or any(key[0][i] >= i or key[0][i] < 0 for i in range(1, len(key[0]))):
            raise ValueError("parents must describe a tree in topological order: parents[0] < 0, 0 <= parents[i] < i")
        t = _parents_cache[key] = torch.tensor(key[0], dtype=torch.int32, device=device)
    return t


def joint_transforms_hip(poses, joints_rest, parents=SMPL_PARENTS, post=None):
    """poses [B, J*3] (or [B,J,3]) fp32 on the GPU, joints_rest [J,3] -> A [B,J,4,4]; ``post`` [J,4,4]: optional per-joint
    right factor (inv(A_t2cano), sings_hybrid.py:398-399).  Differentiable w.r.t. poses and joints_rest.  J <= 64."""
    J = int(joints_rest.shape[0])
    return _JointTransforms.apply(poses.reshape(poses.shape[0], J, 3), joints_rest, _parents_tensor(parents, poses.device), post)


def joint_transforms(pose, joints_rest, parents=SMPL_PARENTS):
    """pose [J*3] or [J,3] axis-angle, joints_rest [J,3] -> A [J,4,4] (relative-to-rest joint transforms).  fp32 tensors on
    the GPU take the HIP kernel (one launch instead of ~100); fp64 / CPU tensors the same math in torch."""
    J = joints_rest.shape[0]
    if pose.is_cuda and pose.dtype == torch.float32 and joints_rest.dtype == torch.float32 and J <= 64:
        return joint_transforms_hip(pose.reshape(1, J * 3), joints_rest, parents)[0]
    return _joint_transforms_torch(pose, joints_rest, parents)


def _joint_transforms_torch(pose, joints_rest, parents=SMPL_PARENTS):
    J = joints_rest.shape[0]
    R = rodrigues(pose.reshape(J, 3))
    rel = joints_rest.clone()
    par = torch.as_tensor(parents[1:], device=joints_rest.device, dtype=torch.long)
    rel[1:] = joints_rest[1:] - joints_rest[par]
    T = torch.zeros(J, 4, 4, dtype=pose.dtype, device=pose.device)
    T[:, :3, :3] = R
    T[:, :3, 3] = rel
    T[:, 3, 3] = 1
    chain = [T[0]]
    for i in range(1, J):
        chain.append(chain[parents[i]] @ T[i])
    G = torch.stack(chain, 0)
    jh = torch.cat([joints_rest, torch.zeros_like(joints_rest[:, :1])], 1)[..., None]
    corr = G @ jh
    return G - torch.cat([torch.zeros(J, 4, 3, dtype=G.dtype, device=G.device), corr], -1)


def cano_to_pose(A_t2pose, A_t2cano):
    """A_cano2pose = A_t2pose @ inverse(A_t2cano)  (sings_hybrid.py:398-399, get_canonical_verts :578-596)."""
    return A_t2pose @ torch.inverse(A_t2cano)


def rotation_6d_to_matrix(d6):
    """sings/rec/utils/geometry/rotations.py:545-566 (call site sings_hybrid.py:356-357, every Gaussian): the HIP kernel of
    sings_amd.rotations (one launch each way); kept here under its round-1 name."""
    from .rotations import rotation_6d_to_matrix as _hip
    return _hip(d6)
