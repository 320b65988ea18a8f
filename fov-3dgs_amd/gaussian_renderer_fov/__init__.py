"""render() for the foveated rasterizer (reference: fov3dgs/gaussian_renderer_fov/__init__.py:19-105)."""
import contextlib
import math
import weakref

import torch

from ..diff_gaussian_rasterization_fov_pcheck_obb import GaussianRasterizationSettings, GaussianRasterizer
from .. import _native
from ..rasterizer import PackedModel, _forward_begin, pack_model, serial_frames, zero_points_like


class _PackState:
    __slots__ = ("refs", "versions", "ptrs", "packed")

    def __init__(self, tensors):
        self.refs = [weakref.ref(t) for t in tensors]
        self.versions = [t._version for t in tensors]
        self.ptrs = [t.data_ptr() for t in tensors]  # `t.data = other` keeps the object and the version, not the storage
        self.packed = None

    def matches(self, tensors):
        return len(self.refs) == len(tensors) and all(
            r() is t and v == t._version and p == t.data_ptr() for r, v, p, t in zip(self.refs, self.versions, self.ptrs, tensors))


def invalidate_packed(pc):
    """Drop the packed copy render(packed="auto") cached on the model `pc` (after a write the autograd version
    counters cannot see: `.data` writes, raw-pointer kernels, DLPack aliases)."""
    if getattr(pc, "_fovraster_pack_state", None) is not None:
        pc._fovraster_pack_state = None


def _auto_packed(pc, means3D, scales, rotations, opacity, shs_rest, shs_dcs, highest_levels):
    """The packed layout of a static model (rasterizer.pack_model), made once and kept on the model object.
    A model counts as static when a call hands over the very same tensor objects, unmodified (autograd version
    counters), as the call before it: the second such call packs, later ones reuse. Models whose getters build new
    tensors on every call (activations evaluated per call) never match and render from the ordinary tensors."""
    tensors = (means3D, scales, rotations, opacity, shs_rest, shs_dcs, highest_levels)
    if any(t is None or not t.is_cuda for t in tensors) or (torch.is_grad_enabled() and any(t.requires_grad for t in tensors)):
        return None
    st = getattr(pc, "_fovraster_pack_state", None)
    if st is not None and st.matches(tensors):
        if st.packed is None:
            st.packed = pack_model(means3D, scales, rotations, opacity, shs=shs_rest, shs_dcs=shs_dcs,
                                   highest_levels=highest_levels)
        return st.packed
    try:
        pc._fovraster_pack_state = _PackState(tensors)
    except AttributeError:
        pass
    return None


def render(viewpoint_camera, pc, bg_color: torch.Tensor, scaling_modifier=1.0, alpha=None, gazeArray=None,
           blending=None, starter=None, ender=None, highest_levels=None, shs_dcs=None, opacities=None, packed=None):
    """Render the scene for one gaze. Background tensor (bg_color) must be on the GPU.
    packed (extension, opt-in; the image is bit-identical either way): None (default) = render from the ordinary tensors,
    exactly the reference's interface; a rasterizer.PackedModel of this model made by pack_model(); or "auto" = made and
    cached here once the model is seen to be static (_auto_packed: same tensor objects, same storage address, same
    autograd version on two consecutive calls). "auto" costs 336 B per Gaussian (2 GB at 6 M) and cannot see writes that
    bypass the version counter (`t.data.copy_()`, raw-pointer kernels, DLPack aliases): call invalidate_packed(pc) after
    such a write."""
    xyz = pc.get_xyz
    if torch.is_grad_enabled():
        screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=xyz.device) + 0
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
        debug=False,
    )
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)

    means3D = xyz
    means2D = screenspace_points
    act = getattr(pc, "get_activated", None)  # extension: the three activations as one fused pass (activations.py)
    if act is not None:
        scales, rotations, opacity = act
        if opacities is not None:
            opacity = opacities
    else:
        opacity = pc.get_opacity if opacities is None else opacities
        scales = pc.get_scaling
        rotations = pc.get_rotation
    shs_rest = pc.get_rest_features

    if isinstance(packed, str):
        packed = _auto_packed(pc, means3D, scales, rotations, opacity, shs_rest, shs_dcs, highest_levels)
    if starter is not None:
        starter.record()
    # (successive inference calls overlap on the GPU -- rasterizer.OVERLAP_SUCCESSIVE_FRAMES; a caller that brackets the call with its
    # own events measures the call's own kernels: that call runs on the caller's stream alone)
    with (serial_frames() if (starter is not None or ender is not None) else contextlib.nullcontext()):
        rendered_image, radii = rasterizer(
            means3D=means3D, means2D=means2D, shs_rest=shs_rest, colors_precomp=None, opacities=opacity, scales=scales,
            rotations=rotations, cov3D_precomp=None, shs_dcs=shs_dcs, highest_levels=highest_levels,
            gazeArray=gazeArray, alpha=alpha, blending=blending, packed=packed)
    if ender is not None:
        ender.record()

    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii}


class PendingRender:
    """A foveated frame between the two halves of its native call (render_begin): finish() -> the dict render() returns."""

    def __init__(self, frame, points):
        self._frame, self._points = frame, points

    def finish(self):
        frame = self._frame
        res = frame.finish()
        radii = res[2]
        with torch.cuda.device(frame.device), torch.cuda.stream(frame.stream):  # `radii` is written on the frame's stream
            visible = radii > 0
        return {"render": res[1], "viewspace_points": self._points, "visibility_filter": visible, "radii": radii}


def render_begin(viewpoint_camera, pc, bg_color: torch.Tensor, scaling_modifier=1.0, alpha=None, gazeArray=None,
                 blending=None, highest_levels=None, shs_dcs=None, opacities=None, packed=None, stream=None):
    """Throughput mode (extension, inference only): enqueue the HEAD of a foveated frame -- tile levels, cull pass, projection,
    tile counts, tile scan -- on `stream` (default: the current one) and return without waiting for its instance count;
    PendingRender.finish() waits for the count and enqueues emission, sort, colours and blend. A host that alternates two
    streams -- begin(n + 1) before finish(n) -- keeps two frames in flight: the latency-bound head of one runs beside the sort
    and the blend of the other (csrc/api.hip). Every stream has its own workspaces; the image of a frame is valid once its
    stream has reached the end of finish()'s work (synchronise the stream, or make the consumer's stream wait for it). Same
    arguments and result as render(). The model's and the camera's tensors must be complete on `stream` (tensors made on another
    stream: make `stream` wait for it first); the copies this package caches itself -- the packed model of packed="auto", the
    contiguous copies of transposed camera matrices -- are handed from stream to stream with events (rasterizer._Produced) or
    kept per stream."""
    with torch.no_grad(), torch.cuda.stream(stream if stream is not None else torch.cuda.current_stream(pc.get_xyz.device)):
        xyz = pc.get_xyz
        rs = GaussianRasterizationSettings(
            image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
            tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg_color,
            scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
            projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree,
            campos=viewpoint_camera.camera_center, prefiltered=False, debug=False)
        act = getattr(pc, "get_activated", None)
        if act is not None:
            scales, rotations, opacity = act
            if opacities is not None:
                opacity = opacities
        else:
            opacity = pc.get_opacity if opacities is None else opacities
            scales, rotations = pc.get_scaling, pc.get_rotation
        shs_rest = pc.get_rest_features
        if isinstance(packed, str):
            packed = _auto_packed(pc, xyz, scales, rotations, opacity, shs_rest, shs_dcs, highest_levels)
        if gazeArray is None:
            raise Exception("gazeArray is required by the foveated rasterizer")
        gaze = gazeArray.detach().flatten().tolist() if isinstance(gazeArray, torch.Tensor) else list(gazeArray)
        empty = torch.Tensor([])
        frame = _forward_begin(_native.VARIANT_FOV_PCHECK_OBB, rs, xyz, shs_rest, empty, opacity, scales, rotations, empty,
                               shs_dcs, highest_levels, (float(gaze[0]), float(gaze[1])), float(alpha), persistent=True,
                               packed=packed)
        return PendingRender(frame, zero_points_like(xyz))
