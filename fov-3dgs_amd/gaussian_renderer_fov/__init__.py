"""render() for the foveated rasterizer (reference: fov3dgs/gaussian_renderer_fov/__init__.py:19-105)."""
import math

import torch

from ..diff_gaussian_rasterization_fov_pcheck_obb import GaussianRasterizationSettings, GaussianRasterizer


def render(viewpoint_camera, pc, bg_color: torch.Tensor, scaling_modifier=1.0, alpha=None, gazeArray=None,
           blending=None, starter=None, ender=None, highest_levels=None, shs_dcs=None, opacities=None, packed=None):
    """Render the scene for one gaze. Background tensor (bg_color) must be on the GPU.
    packed (extension): a rasterizer.PackedModel of this (static) model made by pack_model(); same image, faster binning."""
    xyz = pc.get_xyz
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=xyz.device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass

    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx,
        tanfovy=tanfovy,
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

    means3D = xyz
    means2D = screenspace_points
    opacity = pc.get_opacity if opacities is None else opacities
    scales = pc.get_scaling
    rotations = pc.get_rotation
    shs_rest = pc.get_rest_features

    if starter is not None:
        starter.record()
    rendered_image, radii = rasterizer(
        means3D=means3D, means2D=means2D, shs_rest=shs_rest, colors_precomp=None, opacities=opacity, scales=scales,
        rotations=rotations, cov3D_precomp=None, shs_dcs=shs_dcs, highest_levels=highest_levels,
        gazeArray=gazeArray, alpha=alpha, blending=blending, packed=packed)
    if ender is not None:
        ender.record()

    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii}
