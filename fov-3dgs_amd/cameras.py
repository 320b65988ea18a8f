"""Host-side camera record consumed by render().

Mirrors the attributes the reference's render() entry points read from a camera
(reference: fov3dgs/scene/cameras.py:48-57 ``Camera`` and :60-72 ``MiniCam``;
matrix conventions fov3dgs/utils/graphics_utils.py:38-71): ``world_view_transform``
is the world-to-camera matrix TRANSPOSED, ``full_proj_transform`` = that times the
transposed OpenGL-style projection, ``camera_center`` = row 3 of its inverse.
"""
import math

import numpy as np
import torch


def world_to_view(R, t, translate=(0.0, 0.0, 0.0), scale=1.0):
    """W2C 4x4 (float32) from a camera-to-world rotation R and W2C translation t."""
    Rt = np.zeros((4, 4), dtype=np.float64)
    Rt[:3, :3] = np.asarray(R, dtype=np.float64).T
    Rt[:3, 3] = np.asarray(t, dtype=np.float64)
    Rt[3, 3] = 1.0
    c2w = np.linalg.inv(Rt)
    c2w[:3, 3] = (c2w[:3, 3] + np.asarray(translate, dtype=np.float64)) * scale
    return np.linalg.inv(c2w).astype(np.float32)


def projection_matrix(znear, zfar, fovx, fovy):
    """Perspective matrix with z mapped to [0,1] and w = +z (float32 torch tensor)."""
    tx, ty = math.tan(fovx / 2), math.tan(fovy / 2)
    top, right = ty * znear, tx * znear
    bottom, left = -top, -right
    Pm = torch.zeros(4, 4)
    Pm[0, 0] = 2.0 * znear / (right - left)
    Pm[1, 1] = 2.0 * znear / (top - bottom)
    Pm[0, 2] = (right + left) / (right - left)
    Pm[1, 2] = (top + bottom) / (top - bottom)
    Pm[3, 2] = 1.0
    Pm[2, 2] = zfar / (zfar - znear)
    Pm[2, 3] = -(zfar * znear) / (zfar - znear)
    return Pm


def look_at(eye, target, down=(0.0, 1.0, 0.0)):
    """Camera-to-world rotation R and W2C translation t for a +z-forward, y-down camera."""
    eye = np.asarray(eye, dtype=np.float64)
    f = np.asarray(target, dtype=np.float64) - eye
    f /= np.linalg.norm(f)
    x = np.cross(np.asarray(down, dtype=np.float64), f)
    x /= np.linalg.norm(x)
    y = np.cross(f, x)
    R = np.stack([x, y, f], axis=1)
    t = -R.T @ eye
    return R, t


class MiniCam:
    """The fields render() needs; all tensors live on `device`."""

    def __init__(self, R, t, FoVx, FoVy, width, height, znear=0.01, zfar=100.0, device="cpu"):
        self.image_width, self.image_height = int(width), int(height)
        self.FoVx, self.FoVy = float(FoVx), float(FoVy)
        self.znear, self.zfar = znear, zfar
        wvt = torch.tensor(world_to_view(R, t)).transpose(0, 1)
        proj = projection_matrix(znear, zfar, self.FoVx, self.FoVy).transpose(0, 1)
        self.world_view_transform = wvt.to(device)
        self.projection_matrix = proj.to(device)
        self.full_proj_transform = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).to(device)
        self.camera_center = wvt.inverse()[3, :3].to(device)

    def to(self, device):
        for k in ("world_view_transform", "projection_matrix", "full_proj_transform", "camera_center"):
            setattr(self, k, getattr(self, k).to(device))
        return self
