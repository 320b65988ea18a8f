"""render() of the shared-model foveated baseline (reference: fov3dgs/gaussian_renderer_fov_naive/__init__.py:19-110):
the plain model (one SH set, one opacity per Gaussian) rendered with the foveated extension's tile levels and
per-Gaussian highest levels."""
import math

import torch

from ..diff_gaussian_rasterization_naive_pcheck_obb import GaussianRasterizationSettings, GaussianRasterizer
from ..rasterizer import zero_points_like


def render(viewpoint_camera, pc, bg_color: torch.Tensor, scaling_modifier=1.0, alpha=None, gazeArray=None, blending=None,
           starter=None, ender=None, highest_levels=None):
    """Render the scene for one gaze. Background tensor (bg_color) must be on the GPU."""
    xyz = pc.get_xyz
    if torch.is_grad_enabled():
        screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=xyz.device) + 0
        try:
            screenspace_points.retain_grad()
        except Exception:
            pass
    else:
        screenspace_points = zero_points_like(xyz)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5),
        tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=bg_color,
        scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center,
        prefiltered=False,
        debug=False,
    )
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)
    act = getattr(pc, "get_activated", None)  # extension: the three activations as one fused pass (activations.py)
    if act is not None:
        scales, rotations, opacity = act
    else:
        opacity, scales, rotations = pc.get_opacity, pc.get_scaling, pc.get_rotation
    shs = pc.get_features
    if starter is not None:
        starter.record()
    rendered_image, radii = rasterizer(
        means3D=xyz, means2D=screenspace_points, shs=shs, colors_precomp=None, opacities=opacity, scales=scales,
        rotations=rotations, cov3D_precomp=None, highest_levels=highest_levels, gazeArray=gazeArray, alpha=alpha,
        blending=blending)
    if ender is not None:
        ender.record()
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii}
