"""render() for the non-foveated rasterizers (reference: fov3dgs/gaussian_renderer/__init__.py:19-147).

Same signature and result dict; `pc` is anything exposing the GaussianModel getters
(get_xyz, get_opacity, get_scaling, get_rotation, get_features, get_features_detach_rest,
active_sh_degree), `viewpoint_camera` anything with the Camera/MiniCam fields.
"""
import math

import torch

from ..gaussian_wrapper import get_gs_rasterizer
from ..rasterizer import GaussianRasterizationSettings, zero_points_leaf, zero_points_like


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, masking=False,
           starter=None, ender=None, cuda_type="", loss_map=None, packed=None):
    """Render the scene. Background tensor (bg_color) must be on the GPU.
    packed (extension): a rasterizer.PackedModel of this (static) model made by pack_model(); same image, faster binning."""
    xyz = pc.get_xyz
    # zero tensor that makes autograd return the gradient of the 2D (screen-space) means
    if torch.is_grad_enabled():
        screenspace_points = zero_points_leaf(xyz)
        try:
            screenspace_points.retain_grad()
        except Exception:
            pass
    else:
        screenspace_points = zero_points_like(xyz)

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
        debug=bool(getattr(pipe, "debug", False)),
    )
    rasterizer = get_gs_rasterizer(cuda_type, raster_settings)

    means3D = xyz
    means2D = screenspace_points
    # extensions of this package, picked up when the model offers them: the RAW parameters (the rasterizer applies exp /
    # normalize / sigmoid itself and returns the gradients w.r.t. them: no pass over all P Gaussians on either side of
    # the step); else the three activations as one fused pass (activations.py); else the reference's three getters
    raw = getattr(pc, "get_raw_activation_params", None) if (packed is None and not masking) else None
    act = getattr(pc, "get_activated", None) if raw is None else None
    if raw is not None:
        scales, rotations, opacity = raw
    elif act is not None:
        scales, rotations, opacity = act
    else:
        opacity = pc.get_opacity
        scales = pc.get_scaling
        rotations = pc.get_rotation
    # models that expose their SH tensors separately (get_features_split, an extension of this package) skip the
    # concatenation; any other model goes through the reference's get_features
    if hasattr(pc, "get_features_split"):
        shs = pc.get_features_split_detach_rest if masking else pc.get_features_split
    else:
        shs = pc.get_features_detach_rest if masking else pc.get_features
    if masking:
        scales, means3D, rotations = scales.detach(), means3D.detach(), rotations.detach()

    if starter is not None:
        starter.record()
    extra = {} if packed is None else {"packed": packed}
    if raw is not None:
        extra["raw_activations"] = True
    # extension: gradients as row-sparse tensors (rasterizer.py; only meaningful when every rasterizer input is a leaf parameter,
    # i.e. with the raw parameters and split SH storage)
    if getattr(pc, "row_sparse_grads", False) and raw is not None and hasattr(pc, "get_features_split") and torch.is_grad_enabled():
        extra["row_sparse"] = True
    if cuda_type == "pcheck_obb_loss_weighted_max_count":
        out = rasterizer(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=None, opacities=opacity,
                         scales=scales, rotations=rotations, cov3D_precomp=None, loss_map=loss_map, **extra)
    else:
        out = rasterizer(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=None, opacities=opacity,
                         scales=scales, rotations=rotations, cov3D_precomp=None, **extra)
    if ender is not None:
        ender.record()

    result = {"render": out[0], "viewspace_points": screenspace_points, "visibility_filter": out[1] > 0,
              "radii": out[1]}
    if len(out) == 4:
        result["gs_count"], result["contribs"] = out[2], out[3]
    return result
