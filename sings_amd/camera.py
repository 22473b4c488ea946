"""Camera matrices in the layout the render path consumes.

Mirrors the reference's camera producers (sings/rec/utils/graphics.py:65-85
``get_projection_matrix``, :50-62 ``get_projection_matrix_center``;
sings/rec/datasets/Customdataset.py:96-157 ``init_camera``): ``world_view_transform`` is
the TRANSPOSE of the 4x4 extrinsic, ``full_proj_transform = world_view @ P^T`` and
``camera_center = inverse(world_view)[3, :3]``, all fp32 row-major tensors whose flat
memory is the column-major matrix the rasterizer kernels index.
"""
import math

import numpy as np


def focal2fov(focal, pixels):
    return 2.0 * math.atan(pixels / (2.0 * focal))


def fov2focal(fov, pixels):
    return pixels / (2.0 * math.tan(fov / 2.0))


def get_projection_matrix(znear, zfar, fovX, fovY):
    """graphics.py:65-85 (fp32 element arithmetic, as torch.zeros(4,4) assignment does)."""
    tanHalfFovY = math.tan(fovY / 2)
    tanHalfFovX = math.tan(fovX / 2)
    top = tanHalfFovY * znear
    bottom = -top
    right = tanHalfFovX * znear
    left = -right
    P = np.zeros((4, 4), np.float32)
    z_sign = 1.0
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = z_sign
    P[2, 2] = z_sign * zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def get_projection_matrix_center(znear, zfar, fx, fy, cx, cy, width, height):
    """graphics.py:50-62 (non-centred principal point)."""
    P = np.zeros((4, 4), np.float32)
    cx = width - cx
    z_sign = 1.0
    P[0, 0] = 2.0 * fx / width
    P[1, 1] = 2.0 * fy / height
    P[0, 2] = 1.0 - 2.0 * cx / width
    P[1, 2] = 2.0 * cy / height - 1.0
    P[3, 2] = z_sign
    P[2, 2] = z_sign * zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def make_camera(extrinsic, fx, fy, cx, cy, width, height, znear=0.01, zfar=100.0):
    """Camera dict with the keys ``render`` reads (gs_renderer_single.py:59-82), built the
    way Customdataset.init_camera does (:96-157): centred principal point -> fov branch."""
    extrinsic = np.asarray(extrinsic, np.float32)
    fovx = focal2fov(fx, width)
    fovy = focal2fov(fy, height)
    wvt = np.ascontiguousarray(extrinsic.T).astype(np.float32)
    if abs(cx - width / 2) < 1e-6 and abs(cy - height / 2) < 1e-6:
        P = get_projection_matrix(znear, zfar, fovx, fovy)
    else:
        P = get_projection_matrix_center(znear, zfar, fx, fy, cx, cy, width, height)
    full = (wvt @ P.T).astype(np.float32)
    center = np.linalg.inv(wvt)[3, :3].astype(np.float32)
    return dict(fovx=fovx, fovy=fovy, image_height=int(height), image_width=int(width),
                world_view_transform=wvt, full_proj_transform=full, camera_center=center)


def _camera_datum(world_view_transform, img_hw, fov, znear=0.01, zfar=100.0):
    import torch
    h, w = img_hw
    fx = fov2focal(fov, h)
    cam_int = torch.eye(3)
    cam_int[0, 0] = fx; cam_int[1, 1] = fx; cam_int[0, 2] = w / 2; cam_int[1, 2] = h / 2
    P = torch.from_numpy(get_projection_matrix(znear, zfar, fov, fov)).float().transpose(0, 1)
    return {"fovx": fov, "fovy": fov, "image_height": h, "image_width": w, "world_view_transform": world_view_transform,
            "full_proj_transform": world_view_transform @ P, "camera_center": world_view_transform.inverse()[3, :3],
            "cam_int": cam_int, "cam_ext": world_view_transform, "near": znear, "far": zfar}


def get_static_camera(img_size=512, fov=0.4, device="cuda"):
    """sings/rec/datasets/utils.py:19-57: identity view, square image."""
    import torch
    d = _camera_datum(torch.eye(4), (img_size, img_size), fov)
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in d.items()}


def get_rotating_camera(img_size=512, fov=0.4, dist=5.0, device="cuda", nframes=40, angle_limit=2 * math.pi):
    """sings/rec/datasets/utils.py:60-120: nframes cameras orbiting the origin at height -0.25, distance `dist` (the
    reference's ``rot_z`` is a rotation about the y axis; y and z of the camera frame are flipped)."""
    import math
    import torch
    if angle_limit is None:
        angle_limit = 2 * math.pi
    hw = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
    out = []
    for azim in torch.linspace(0, angle_limit, nframes):
        rot = lambda a: torch.tensor([[torch.cos(a), 0, torch.sin(a)], [0, 1, 0], [-torch.sin(a), 0, torch.cos(a)]])
        t = (rot(-azim) @ torch.tensor([[0., -0.25, dist]]).T).T
        R = rot(azim)[None]
        R[:, 1:3] *= -1
        Rt = torch.eye(4)
        Rt[:3, :3] = R[0].T
        Rt[:3, 3] = t[0].squeeze()
        d = _camera_datum(Rt.inverse().T, hw, fov)
        out.append({k: (v.to(device) if torch.is_tensor(v) else v) for k, v in d.items()})
    return out
