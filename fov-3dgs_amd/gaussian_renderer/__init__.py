"""render() for the non-foveated rasterizers (reference: fov3dgs/gaussian_renderer/__init__.py:19-147).

Same signature and result dict; `pc` is anything exposing the GaussianModel getters
(get_xyz, get_opacity, get_scaling, get_rotation, get_features, get_features_detach_rest,
active_sh_degree), `viewpoint_camera` anything with the Camera/MiniCam fields.
"""
import inspect
import math

import torch

from ..gaussian_wrapper import get_gs_rasterizer
from ..rasterizer import GaussianRasterizationSettings, zero_points_leaf, zero_points_like


# The reference's GaussianModel keeps its raw parameter tensors and its three activation functions as plain attributes
# (scene/gaussian_model.py:33-41: scaling_activation = torch.exp, opacity_activation = torch.sigmoid, rotation_activation =
# torch.nn.functional.normalize; :43-50: _features_dc, _features_rest, _scaling, _rotation, _opacity) and its getters are one-line
# expressions of them (:200-240). render() recognises such a model and then hands the rasterizer what the getters would have been
# computed FROM: the two SH tensors as they are stored (get_features is their torch.cat: 1.15 GB written and read back per step at
# 6 M Gaussians, and split again by autograd) and the raw parameters (the kernels apply exp / normalize / sigmoid themselves and return
# the gradients w.r.t. the raw tensors) -- 4.7 -> 2.3 ms per training step on the S-6M cloud with no change to the model class. Same
# values up to the last bit of the device's exp / sigmoid. Set to False to go through the getters only.
#
# Recognition is by what the getters DO, not by what the object carries (a duck-typed model or a subclass that keeps the attributes
# but overrides a getter -- opacity times a learned mask, clamped scales, another feature concat -- must get ITS getters):
#   1. the three activation attributes are exactly torch.exp / torch.sigmoid / torch.nn.functional.normalize and the raw tensors have
#      the reference's shapes;
#   2. every getter render() would have called resolves, through the class's MRO, to a property whose code is the reference's
#      one-liner: it touches exactly the names that one-liner touches and no constants of its own (_getter_fingerprint_ok, per class);
#   3. once per class, the getters' outputs are compared with activation(raw) / torch.cat(dc, rest) bit for bit (_self_check).
# Anything else goes through its getters.
FAST_REFERENCE_MODEL = True

# names each reference getter touches (scene/gaussian_model.py:200-240) and the constants it may hold
_REFERENCE_GETTERS = {
    "get_xyz": ({"_xyz"}, (None,)),
    "get_scaling": ({"scaling_activation", "_scaling"}, (None,)),
    "get_rotation": ({"rotation_activation", "_rotation"}, (None,)),
    "get_opacity": ({"opacity_activation", "_opacity"}, (None,)),
    "get_features": ({"_features_dc", "_features_rest", "torch", "cat"}, (None, 1, ("dim",))),
    "get_features_detach_rest": ({"_features_dc", "_features_rest", "detach", "torch", "cat"}, (None, 1, ("dim",))),
}
_class_verdict = {}   # class -> bool: fingerprints ok (None entry never stored)
_class_checked = {}   # class -> bool: numeric self-check passed


def _getter_fingerprint_ok(cls):
    """True when every getter of `cls` that render() replaces is, as found through the MRO, a property whose function touches exactly
    the names of the reference's one-line getter and carries no constant of its own (a factor, a clamp bound, another dim)."""
    hit = _class_verdict.get(cls)
    if hit is not None:
        return hit
    ok = True
    for name, (names, consts) in _REFERENCE_GETTERS.items():
        attr = inspect.getattr_static(cls, name, None)
        code = getattr(getattr(attr, "fget", None), "__code__", None)
        if not isinstance(attr, property) or code is None:
            ok = False
            break
        own = [c for c in code.co_consts if c != attr.fget.__doc__]   # (a docstring is the function's first constant)
        if set(code.co_names) != names or code.co_argcount != 1 or any(inspect.iscode(c) or c not in consts for c in own):
            ok = False
            break
    _class_verdict[cls] = ok
    return ok


def _self_check(pc, f):
    """Once per class: what the getters return IS activation(raw) / the concatenation, to the last bit."""
    cls = type(pc)
    hit = _class_checked.get(cls)
    if hit is not None:
        return hit
    with torch.no_grad():
        ok = (pc.get_xyz is pc._xyz or torch.equal(pc.get_xyz, pc._xyz)) and torch.equal(pc.get_scaling, torch.exp(f[0])) \
            and torch.equal(pc.get_rotation, torch.nn.functional.normalize(f[1])) and torch.equal(pc.get_opacity, torch.sigmoid(f[2])) \
            and torch.equal(pc.get_features, torch.cat((f[3], f[4]), dim=1))
    _class_checked[cls] = bool(ok)
    return _class_checked[cls]


def _reference_model_fields(pc):
    """-> (raw scaling, raw rotation, raw opacity, features_dc, features_rest) of a model that IS the reference's GaussianModel as far as
    render() can tell (see above), or None when `pc` is anything else (a subclass that overrides a getter, other activations, a
    wrapper exposing only getters, ...)."""
    if not FAST_REFERENCE_MODEL:
        return None
    try:
        if not (pc.scaling_activation is torch.exp and pc.opacity_activation is torch.sigmoid
                and pc.rotation_activation is torch.nn.functional.normalize):
            return None
        f = (pc._scaling, pc._rotation, pc._opacity, pc._features_dc, pc._features_rest)
        xyz = pc._xyz
    except AttributeError:
        return None
    if not _getter_fingerprint_ok(type(pc)):
        return None
    P = xyz.shape[0]
    if not all(isinstance(t, torch.Tensor) and t.is_cuda and t.dim() >= 2 and t.shape[0] == P for t in f):
        return None
    if tuple(f[0].shape) != (P, 3) or tuple(f[1].shape) != (P, 4) or f[2].numel() != P or f[3].shape[1] != 1 or f[3].shape[2:] != f[4].shape[2:]:
        return None
    if not _self_check(pc, f):
        return None
    return f


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, masking=False,
           starter=None, ender=None, cuda_type="", loss_map=None, packed=None, want_stats=True):
    """Render the scene. Background tensor (bg_color) must be on the GPU.
    packed (extension): a rasterizer.PackedModel of this (static) model made by pack_model(); same image, faster binning.
    want_stats (extension, opt-in): False = the caller does not read result["gs_count"] / ["contribs"] of the pcheck_obb_sum
    rasterizer (eff_finetune.py:107-108 drops them every step): the blend skips the per-Gaussian statistics and the two keys
    are absent from the result; image, radii and gradients are unchanged."""
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
    ref_fields = _reference_model_fields(pc) if (raw is None and act is None and not hasattr(pc, "get_features_split")) else None
    if ref_fields is not None and packed is None and not masking:
        raw = ref_fields[:3]
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
    elif ref_fields is not None:
        shs = (ref_fields[3], ref_fields[4].detach() if masking else ref_fields[4])
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
    if not want_stats and cuda_type == "pcheck_obb_sum":
        extra["want_stats"] = False
    if getattr(pc, "row_sparse_grads", False) and raw is not None and isinstance(shs, tuple) and torch.is_grad_enabled():
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
