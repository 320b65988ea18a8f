"""Shared test helpers: seeded scenes -> oracle input dicts / torch tensors."""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import fov3dgs_amd  # noqa: E402,F401  (import shim for the hyphenated package dir)
from fov3dgs_amd import synthetic as syn  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
FOV_VARIANT = "fov_pcheck_obb"


def cam_dict(cam, bg=(0.0, 0.0, 0.0), sh_degree=3, gaze=(0.5, 0.5), alpha=0.05, scale_modifier=1.0):
    return dict(image_width=cam.image_width, image_height=cam.image_height,
                tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
                bg=np.asarray(bg, np.float32), viewmatrix=cam.world_view_transform.cpu().numpy(),
                projmatrix=cam.full_proj_transform.cpu().numpy(), campos=cam.camera_center.cpu().numpy(),
                sh_degree=sh_degree, gaze=gaze, alpha=alpha, scale_modifier=scale_modifier, prefiltered=False)


def scene_dict(cloud, variant, fov=None):
    """Oracle input dict from a GaussianCloud (activations applied, as render() does)."""
    with torch.no_grad():
        d = dict(means3D=cloud.get_xyz.cpu().numpy(), scales=cloud.get_scaling.cpu().numpy(),
                 rotations=cloud.get_rotation.cpu().numpy())
        if variant == FOV_VARIANT:
            highest, shs_dcs, opac = fov
            d.update(shs=cloud.get_rest_features.cpu().numpy(), opacities=opac.cpu().numpy(),
                     shs_dcs=shs_dcs.cpu().numpy(), highest_levels=highest.cpu().numpy())
        else:
            d.update(shs=cloud.get_features.cpu().numpy(), opacities=cloud.get_opacity.cpu().numpy())
    return d


def small_cloud(P=3000, seed=3, big_fraction=0.1):
    """A small cloud with a tail of large splats (multi-tile rects, OBB culls, long lists)."""
    cloud = syn.scene_1k(P=P, seed=seed)
    g = torch.Generator().manual_seed(seed + 100)
    nbig = int(P * big_fraction)
    idx = torch.randperm(P, generator=g)[:nbig]
    cloud._scaling[idx] += 1.2 + 0.5 * torch.rand(nbig, 1, generator=g)
    cloud._scaling[idx, 0] += 1.0  # anisotropic
    # a few behind / near the camera plane to exercise the near cull
    cloud._xyz[idx[: nbig // 4], 2] -= 4.0
    cloud._opacity += 1.0  # denser, so some pixels saturate (T < 1e-4 early stop)
    return cloud


def small_camera(width=200, height=120):
    """Ragged tile grid (200x120 is not a multiple of 16 in either axis)."""
    fovx = math.radians(70.0)
    fx = width / (2 * math.tan(fovx / 2))
    fovy = 2 * math.atan(height / (2 * fx))
    R, t = syn.look_at((0.4, -0.3, -0.5), (0.0, 0.0, 4.0))
    return syn.MiniCam(R, t, fovx, fovy, width, height)


def small_case(variant, P=3000, seed=3, bg=(0.1, 0.2, 0.3), gaze=(0.4, 0.55), alpha=0.05, width=200, height=120):
    cloud = small_cloud(P, seed)
    cam = small_camera(width, height)
    fov = syn.foveation_layers(cloud, seed=seed + 1) if variant == FOV_VARIANT else None
    scene = scene_dict(cloud, variant, fov)
    if variant == "pcheck_obb_loss_weighted_max_count":
        # the callers pass a [3,H,W] map (prune.py:80); only its first plane is read
        scene["loss_map"] = np.random.default_rng(seed + 7).random((3, height, width)).astype(np.float32)
    return scene, cam_dict(cam, bg=bg, gaze=gaze, alpha=alpha)
