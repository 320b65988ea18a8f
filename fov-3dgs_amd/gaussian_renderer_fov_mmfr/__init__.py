"""render() of the multi-model foveated baseline (reference: fov3dgs/gaussian_renderer_fov_mmfr/__init__.py:19-175): one
full model per eccentricity level, every level rendered by its own rasterizer call (which skips the tiles outside its
level band and weights the pixels of two-level tiles), the level images added up."""
import math

import torch

from ..diff_gaussian_rasterization_mmfr_pcheck_obb import GaussianRasterizationSettings, GaussianRasterizer
from ..rasterizer import zero_points_like


def render(viewpoint_camera, bg_color: torch.Tensor, scaling_modifier=1.0, alpha=None, gazeArray=None, blending=None,
           starter=None, ender=None, multi_gs=None, layer_num=None):
    """Render the scene for one gaze from the models multi_gs[0..]. Background tensor (bg_color) must be on the GPU.
    (The reference unrolls exactly four levels; any number of models is accepted here, layer_num is unused there too.)"""
    xyz0 = multi_gs[0].get_xyz
    if torch.is_grad_enabled():
        screenspace_points = torch.zeros_like(xyz0, dtype=xyz0.dtype, requires_grad=True, device=xyz0.device) + 0
        try:
            screenspace_points.retain_grad()
        except Exception:
            pass
    else:
        screenspace_points = zero_points_like(xyz0)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5),
        tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=bg_color,
        scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=multi_gs[0].active_sh_degree,
        campos=viewpoint_camera.camera_center,
        prefiltered=False,
        debug=False,
    )
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)
    inputs = []
    for gs in multi_gs:
        act = getattr(gs, "get_activated", None)
        scales, rotations, opacity = act if act is not None else (gs.get_scaling, gs.get_rotation, gs.get_opacity)
        inputs.append((gs.get_xyz, opacity, scales, rotations, gs.get_features))
    if starter is not None:
        starter.record()
    total, radii = None, None
    for level, (means3D, opacity, scales, rotations, shs) in enumerate(inputs):
        image, radii = rasterizer(means3D=means3D, means2D=screenspace_points, shs=shs, colors_precomp=None, opacities=opacity,
                                  scales=scales, rotations=rotations, cov3D_precomp=None, cur_level=level, gazeArray=gazeArray,
                                  alpha=alpha, blending=blending)
        total = image if total is None else total + image
    if ender is not None:
        ender.record()
    return {"render": total, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}
