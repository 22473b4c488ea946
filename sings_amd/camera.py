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
