"""Host side of the drop-in boundary: the reference's autograd interface over the C ABI.

Mirrors, per variant, the Python wrapper each reference extension ships
(fov3dgs/submodules/<variant>/<pkg>/__init__.py):
  GaussianRasterizationSettings   diff_gaussian_rasterization/__init__.py:157-169 (same 12 fields everywhere)
  GaussianRasterizer              …/__init__.py:171-220; RS :177-226 (4 outputs); RF :203-261 (5 extra inputs)
  _RasterizeGaussians             …/__init__.py:44-155 (argument order, saved tensors, gradient order)
  _C.rasterize_gaussians[_backward] / mark_visible   rasterize_points.cu:35-217 -> fr_forward / fr_backward /
                                  fr_mark_visible through ctypes (fov3dgs_amd/_native.py)

All tensors must live on a ROCm device; there is no CPU path. Output / workspace tensors are
allocated here with torch (the C library never allocates device memory).
"""
import contextlib
import ctypes as C
import os
import threading
import time
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _native


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def cpu_deep_copy_tuple(input_tuple):
    return tuple(item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple)


def _f32(t, device):
    """contiguous fp32 tensor on `device`, or None for the reference's 'empty tensor' placeholder."""
    if t is None or t.numel() == 0:
        return None
    if t.device != device:
        raise RuntimeError(f"fovraster: expected every tensor on {device}, got one on {t.device}")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _ptr(t):
    return None if t is None else t.data_ptr()


_small_copies = {}
SMALL_COPY_CACHE = True  # set to False to copy the per-call camera tensors on every call


def _f32_small(t, device, stream_handle=None):
    """_f32 for the per-call camera tensors (viewmatrix, projmatrix, campos, bg: a few floats each). The reference's cameras
    keep world_view_transform as a TRANSPOSED view (scene/cameras.py:54), so `.contiguous()` is a copy kernel on the stream
    at the head of every frame (~6 us of a 0.7 ms frame, twice); the copy is kept for as long as the caller hands over the
    same tensor object with the same storage, strides and autograd version -- PER STREAM: the copy kernel is enqueued on the
    stream that was current when it was made, so a frame on another stream (render_begin alternates two) makes and keeps its
    own copy instead of reading one that may still be in flight elsewhere. Writes the version counter cannot see (`.data`
    writes, raw-pointer kernels, DLPack / numpy aliases) are caught for CPU tensors by comparing the values (a few floats on
    the host); for a device tensor clone it after such a write, or set SMALL_COPY_CACHE = False."""
    if t is None or t.numel() == 0:
        return None
    if t.device == device and t.dtype == torch.float32 and t.is_contiguous():
        return t
    if t.numel() > 64 or not SMALL_COPY_CACHE:
        return _f32(t, device)
    key = (id(t), torch.cuda.current_stream(device).cuda_stream if stream_handle is None else stream_handle)  # (the lookup costs 6 us of host time)
    ent = _small_copies.get(key)
    sig = (t._version, t.data_ptr(), t.stride(), t.dtype, t.device)
    if ent is not None and ent[0]() is t and ent[1] == sig and (t.is_cuda or torch.equal(ent[3], t)):
        # (a stream handle may be a NEW stream's -- handles of destroyed streams are reused --, which is not ordered behind the copy
        # kernel the old stream ran: until that kernel is known to be over the reader waits for its event)
        if not ent[4][0]:
            if ent[5].query():
                ent[4][0] = True
            else:
                torch.cuda.current_stream(device).wait_event(ent[5])
        return ent[2]
    c = _f32(t, device)
    if len(_small_copies) > 256:
        _small_copies.clear()
    import weakref
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    _small_copies[key] = (weakref.ref(t), sig, c, None if t.is_cuda else t.detach().clone(), [False], ev)
    return c


class _Produced:
    """Cross-stream hand-over of a cached device buffer: the event recorded behind its producer kernels on the stream that made
    it. A consumer on ANOTHER stream waits for that event (once per stream) and tells the caching allocator that the buffer is in
    use there (record_stream), so the buffer is neither read half-written nor recycled under a running kernel."""

    def __init__(self, device, tensors):
        self.stream = torch.cuda.current_stream(device)  # (kept alive: its handle cannot be handed to another stream meanwhile)
        self.event = torch.cuda.Event()
        self.event.record(self.stream)
        self.tensors = [t for t in tensors if t is not None]
        self.complete = False
        self.seen = set()

    def wait_on_current(self, device):
        cur = torch.cuda.current_stream(device)
        if cur.cuda_stream == self.stream.cuda_stream:
            return
        # Stream handles are reused once a stream is destroyed (a host that makes a stream per frame): `seen` only short-cuts the
        # WAIT, and only until the producer's event is known to be over -- after that no stream has anything to wait for. The
        # allocator is told about every consumer stream on every call (a set insertion per tensor).
        if not self.complete:
            if self.event.query():
                self.complete = True
            elif cur.cuda_stream not in self.seen:
                cur.wait_event(self.event)
                self.seen.add(cur.cuda_stream)
        for t in self.tensors:
            t.record_stream(cur)


def _require_gpu(means3D):
    if not means3D.is_cuda:
        raise RuntimeError("fovraster: tensors must live on a ROCm GPU -- the rasterizer is a HIP extension and has no "
                           "CPU fallback (the CPU oracle under oracle/ is test infrastructure only)")


_GRANULE = 32 << 20  # workspace sizes are rounded up so the caching allocator can reuse blocks


class _Workspaces:
    """The three byte buffers the native call sizes through callbacks (reference: resizeFunctional).

    The binning buffer's size changes every frame with the instance count; un-rounded requests make
    PyTorch's caching allocator fall through to hipMalloc (~8 ms each on MI355X), so requests are
    rounded up to 32 MiB granules, and inference calls (no autograd graph) keep one grow-only set of
    buffers per device alive across frames.
    """

    def __init__(self, device):
        self.device = device
        self.buf = [torch.empty(0, dtype=torch.uint8, device=device) for _ in range(3)]
        self.cbs = [_native.RESIZE_FN(self._make(i)) for i in range(3)]

    def _make(self, i):
        def resize(_user, nbytes):
            if self.buf[i].numel() < nbytes:
                # (no headroom on top: the library asks for the binning workspace with a quarter of headroom over the largest
                # frame of the kind itself, include/fovraster.h)
                want = (int(nbytes) + _GRANULE - 1) // _GRANULE * _GRANULE
                self.buf[i] = torch.empty(want, dtype=torch.uint8, device=self.device)
            return self.buf[i].data_ptr()
        return resize


import collections

_persistent_ws = collections.OrderedDict()
PERSISTENT_WS_SETS = 8  # per device: grow-only inference workspace sets kept alive (least recently used goes first); the caller's stream, the three internal streams of successive-frame overlap and a two-streams-in-flight host together use six
_pooled_ws = {}  # device -> idle workspace sets of finished training steps


class _Lease:
    """Keeps one workspace set out of the pool while an autograd graph may still read it in backward();
    the set goes back when the last reference (the autograd context) dies."""

    def __init__(self, ws):
        self.ws = ws

    def __del__(self):
        try:
            _pooled_ws.setdefault(self.ws.device, []).append(self.ws)
        except Exception:  # interpreter shutdown
            pass


def _workspaces_for(device, needs_graph, stream_handle=None):
    """-> (workspaces, lease). Calls without an autograd graph use a persistent grow-only set per (device, current stream, host
    thread) -- the C ABI is thread-compatible and two frames may be in flight on two streams (begin / finish), so neither two
    threads nor two streams ever share buffers; a call is only valid until the next call on the same stream of the same
    thread. At most PERSISTENT_WS_SETS sets per device are kept (a set is ~1.5-2 GB at 6 M Gaussians; stream handles and thread
    idents come and go, a host that makes a stream per frame must not pin one set per stream for ever): the least recently used
    one is dropped -- a frame in flight on it holds its own reference until it has finished, and the caching allocator hands the
    blocks back to the stream they were allocated on. For training a set from a small pool, leased until the graph is gone:
    allocating GBs of fresh buffers every step made the caching allocator fall back to hipMalloc/hipFree every few steps (30-40 ms
    stalls)."""
    if needs_graph:
        idle = _pooled_ws.get(device)
        ws = idle.pop() if idle else _Workspaces(device)
        return ws, _Lease(ws)
    key = (device, torch.cuda.current_stream(device).cuda_stream if stream_handle is None else stream_handle, threading.get_ident())
    ws = _persistent_ws.get(key)
    if ws is None:
        ws = _persistent_ws[key] = _Workspaces(device)
        mine = [k for k in _persistent_ws if k[0] == device]
        for k in mine[:max(0, len(mine) - PERSISTENT_WS_SETS)]:
            del _persistent_ws[k]
    else:
        _persistent_ws.move_to_end(key)
    return ws, None


_zero_rows = {}


def zero_points_like(xyz):
    """All-zero [P,3] tensor for render()'s `viewspace_points` when autograd is off: a stride-0 view of one cached zero
    row instead of the reference's zeros_like(xyz) + 0 (a 72 MB fill and a 144 MB add per frame at 6 M Gaussians that
    only exist to carry a gradient). Same shape, dtype and values; read-only by convention."""
    key = (xyz.device, xyz.dtype)
    row = _zero_rows.get(key)
    if row is None:
        row = _zero_rows[key] = torch.zeros((1, xyz.shape[-1]), dtype=xyz.dtype, device=xyz.device)
    return row.expand(xyz.shape)


_zero_leaves = {}
FRESH_VIEWSPACE_POINTS = bool(int(__import__("os").environ.get("FOVRASTER_FRESH_VIEWSPACE_POINTS", "0")))


def zero_points_leaf(xyz):
    """render()'s `viewspace_points` of a training step: a NEW leaf tensor (requires_grad, its own .grad) of zeros shaped like xyz
    -- the reference's torch.zeros_like(xyz, requires_grad=True) -- over ONE cached zero buffer per (device, dtype, shape): the
    tensor only carries the gradient of the 2D means, the rasterizer never reads or writes its values, and filling 72 MB per step
    at 6 M Gaussians is 16 us at the head of every forward pass. A caller that writes INTO a step's viewspace_points in place
    would leave that in the shared buffer: every leaf shares the buffer's version counter, so the next call sees the write and
    clears the buffer again (the reference's tensor is private to its step; so are the values here, one step late at worst --
    and never read by this package). FOVRASTER_FRESH_VIEWSPACE_POINTS=1 allocates fresh zeros per step instead."""
    if FRESH_VIEWSPACE_POINTS:
        return torch.zeros(xyz.shape, dtype=xyz.dtype, device=xyz.device, requires_grad=True)
    key = (xyz.device, xyz.dtype, tuple(xyz.shape))
    ent = _zero_leaves.get(key)
    if ent is None:
        if len(_zero_leaves) >= 4:
            _zero_leaves.clear()  # (a model that changes size every few steps: densification)
        buf = torch.zeros(xyz.shape, dtype=xyz.dtype, device=xyz.device)
        ent = _zero_leaves[key] = [buf, buf._version]
    buf = ent[0]
    if buf._version != ent[1]:  # somebody wrote into an earlier step's leaf
        with torch.no_grad():
            buf.zero_()
        ent[1] = buf._version
    return buf.detach().requires_grad_(True)


class PackedModel:
    """Packed copies of a STATIC model's rasterizer inputs (include/fovraster.h: packed_geom [P,16] /
    packed_colour [P,64] / packed_cull [P,4]): made once per model with pack_model(), passed to GaussianRasterizer(..., packed=...) next
    to the ordinary tensors, which must hold the same values. Forward-only (ignored by the backward pass); the
    image is bit-identical with and without."""

    def __init__(self, geom, colour, cull):
        self.geom, self.colour, self.cull = geom, colour, cull
        # the k_pack_* kernels run on the stream that is current now: a frame on another stream waits for them (_Produced)
        self.produced = _Produced(geom.device, (geom, colour, cull)) if geom.is_cuda else None

    def check(self, P, dev):
        for t, w in ((self.geom, 16), (self.colour, 64), (self.cull, 4)):
            if t is not None and (t.device != dev or t.dtype != torch.float32 or tuple(t.shape) != (P, w) or not t.is_contiguous()):
                raise RuntimeError(f"packed model does not match: expected float32 [{P}, {w}] on {dev}")


def pack_model(means3D, scales, rotations, opacities, shs=None, shs_rest=None, shs_dcs=None, highest_levels=None):
    """-> PackedModel. opacities [P,1] (or [P,4] per level for the foveated rasterizer, with shs_dcs [P,4,3],
    highest_levels [P,1] and shs = the 15 rest coefficients); shs [P,16,3], or shs [P,1,3] + shs_rest [P,15,3].
    """
    lib = _native.load()
    _require_gpu(means3D)
    dev = means3D.device
    P = means3D.size(0)
    f = lambda t: None if t is None else _f32(t.detach(), dev)
    m, sc, ro, op, sh, rest, dcs, hl = (f(t) for t in (means3D, scales, rotations, opacities, shs, shs_rest, shs_dcs, highest_levels))
    levels = op.numel() // max(P, 1) if P else 1
    geom = torch.empty((P, 16), dtype=torch.float32, device=dev)
    cull = torch.empty((P, 4), dtype=torch.float32, device=dev)
    if sh is None:
        raise RuntimeError("pack_model needs the SH coefficients (the packed layout has no colors_precomp form)")
    colour = None
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = lib.fr_pack_geom(P, _ptr(m), _ptr(sc), _ptr(ro), _ptr(op), int(levels), _ptr(hl), _ptr(geom), stream)
        if rc == 0:
            rc = lib.fr_pack_cull(P, _ptr(m), _ptr(sc), _ptr(ro), _ptr(cull), stream)
        if rc != 0:
            raise RuntimeError(f"fovraster pack_geom / pack_cull failed ({rc}): {_native.last_error()}")
        if sh is not None:
            ncoef = sh.size(1) + (rest.size(1) if rest is not None else 0)
            if ncoef != (15 if dcs is not None else 16):
                raise RuntimeError(f"pack_model needs all 16 SH coefficients, got {ncoef}")
            colour = torch.empty((P, 64), dtype=torch.float32, device=dev)
            rc = lib.fr_pack_colour(P, _ptr(sh), _ptr(rest), _ptr(dcs), _ptr(colour), stream)
            if rc != 0:
                raise RuntimeError(f"fovraster pack_colour failed ({rc}): {_native.last_error()}")
    return PackedModel(geom, colour, cull)


# Set by fov3dgs_amd.profiling.StageTimer while a timed region is active: a ctypes array of
# FR_NUM_STAGE_EVENTS event handles that the next forward call records on its streams.
_stage_events_hook = None
_call_events = None  # profiling.NativeCallTimer: a list that takes (kind, start, stop) torch events around every native fr_forward / fr_backward call
_bwd_events_hook = None  # the same for backward calls: 5 handles (fr_backward_args.stage_events)


class FrameInFlight:
    """A forward call between its two halves (fr_forward_begin / fr_forward_finish, include/fovraster.h): the head of the
    frame is on the stream, finish() waits for the instance count and enqueues the rest. Holds everything the native call
    reads (argument struct, tensors, workspaces) alive until then."""

    def __init__(self, lib, a, keep, color, radii, ws, lease, counts, contribs, handle, device, stream):
        self.lib, self.a, self.keep, self.color, self.radii = lib, a, keep, color, radii
        self.ws, self.lease, self.counts, self.contribs, self.handle = ws, lease, counts, contribs, handle
        self.device, self.stream = device, stream

    def finish(self, on_stream=False):
        """-> (num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer[, gaussians_count, contributions], lease)
        on_stream: the frame's stream IS the current one (the caller says so: saves two context managers)"""
        handle, self.handle = self.handle, None
        if handle is None:
            raise RuntimeError("fovraster: frame already finished")
        # (the binning workspace callback allocates: on the frame's own stream, whatever stream the caller is on by now)
        if on_stream:
            rc = self.lib.fr_forward_finish(handle)
        else:
            with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
                rc = self.lib.fr_forward_finish(handle)
        if rc != 0:
            raise RuntimeError(f"fovraster forward failed ({rc}): {_native.last_error()}")
        ws = self.ws
        out = (int(self.a.num_rendered), self.color, self.radii, ws.buf[0], ws.buf[1], ws.buf[2])
        if self.counts is not None:
            out = out + (self.counts, self.contribs)
        return out + (self.lease,)  # last element: keeps the workspace set reserved (None for the persistent set)

    def __del__(self):
        handle, self.handle = getattr(self, "handle", None), None
        if handle is not None:  # abandoned between the halves: the library must release its handle -- without running the tail
            try:
                self.lib.fr_forward_abandon(handle)
            except Exception:  # interpreter shutdown
                pass


def _forward_begin(variant, rs, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                   shs_dcs=None, highest_levels=None, gaze=(0.5, 0.5), alpha=0.05, persistent=False, loss_map=None,
                   sh_rest=None, packed=None, cur_level=0.0, raw_activations=False, list_consumed=None, no_stats=False, blend_pairs=None,
                   on_stream=None, reuse=None, reuse_key=None):
    """First half of a forward call on the current stream -> FrameInFlight. on_stream: the caller HAS made this stream (and its
    device) current and says so (saves the lookups). persistent=True: the workspaces are the grow-only
    set of this (device, stream, thread) (valid until the next call there); otherwise they stay reserved for as long as the
    `lease` of the result is referenced.
    reuse / reuse_key (the overlapped inference path): `reuse` is a dict the caller keeps per internal stream; when its "key" equals
    reuse_key -- the caller vouches that every tensor, scalar and flag of the call is what it was at the previous call on that stream --
    the argument struct made then is used again with only the gaze and the two output tensors replaced (filling it is 20 checked
    tensor arguments: 40 us of host time on a path where the host is the bottleneck)."""
    lib = _native.load()
    _require_gpu(means3D)
    dev = means3D.device
    P = means3D.size(0)
    H, W = int(rs.image_height), int(rs.image_width)
    stream = torch.cuda.current_stream(dev) if on_stream is None else on_stream  # (looked up once: 6 us of host time a call)
    sh_ = stream.cuda_stream
    if reuse is not None and reuse_key is not None and reuse.get("key") == reuse_key and on_stream is not None:
        a, keep, ws = reuse["a"], reuse["keep"], reuse["ws"]
        color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        a.gaze_x, a.gaze_y = float(gaze[0]), float(gaze[1])
        a.out_color, a.radii = color.data_ptr(), radii.data_ptr()
        a.stage_events = _stage_events_hook() if _stage_events_hook is not None else None
        handle = C.c_void_p()
        rc = lib.fr_forward_begin(C.byref(a), C.byref(handle))
        if rc != 0:
            raise RuntimeError(f"fovraster forward failed ({rc}): {_native.last_error()}")
        return FrameInFlight(lib, a, keep, color, radii, ws, None, None, None, handle, dev, stream)
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    a = _native.ForwardArgs()
    keep = []

    own = []  # the tensors this call MADE (converted copies): what the argument struct must keep alive if it is used again

    def put(name, t, small=False):
        src = t
        t = _f32_small(t, dev, sh_) if small else _f32(t, dev)
        keep.append(t)
        if t is not None and t is not src:
            own.append(t)
        setattr(a, name, _ptr(t))
        return t

    with (contextlib.nullcontext() if on_stream is not None else torch.cuda.device(dev)):
        # both outputs are written in full by the kernels (every pixel by the blend, every radius by the cull pass
        # or the projection kernel; the P == 0 path fills the image itself): no zero-fill kernels at the head of the frame
        color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        ws, lease = _workspaces_for(dev, not persistent, sh_)
        counts = contribs = None
        a.variant = variant
        a.P, a.D = P, int(rs.sh_degree)
        sh_c = put("shs", sh)
        rest_c = put("shs_rest", sh_rest)  # split storage: sh = DC [P,1,3], sh_rest = [P,M-1,3]
        a.M = 0 if sh_c is None else sh_c.size(1) + (0 if rest_c is None else rest_c.size(1))
        a.W, a.H = W, H
        a.prefiltered, a.debug = int(bool(rs.prefiltered)), int(bool(rs.debug))
        a.tanfovx, a.tanfovy = float(rs.tanfovx), float(rs.tanfovy)
        a.scale_modifier = float(rs.scale_modifier)
        a.gaze_x, a.gaze_y, a.alpha = float(gaze[0]), float(gaze[1]), float(alpha)
        a.cur_level = float(cur_level)
        a.raw_activations = int(bool(raw_activations))
        a.stream = sh_
        put("background", rs.bg, small=True)
        put("means3D", means3D)
        put("colors_precomp", colors_precomp)
        put("opacities", opacities)
        put("scales", scales)
        put("rotations", rotations)
        put("cov3D_precomp", cov3Ds_precomp)
        put("viewmatrix", rs.viewmatrix, small=True)
        put("projmatrix", rs.projmatrix, small=True)
        put("campos", rs.campos, small=True)
        put("shs_dcs", shs_dcs)
        put("highest_levels", highest_levels)
        a.out_color, a.radii = color.data_ptr(), radii.data_ptr()
        put("loss_map", loss_map)
        if list_consumed is not None:  # diagnostic: uint32 / int32 [T], entries of every tile's list the blend fetched (fovraster.h)
            tiles = ((W + 15) // 16) * ((H + 15) // 16)
            if list_consumed.device != dev or list_consumed.numel() != tiles or list_consumed.element_size() != 4 or not list_consumed.is_contiguous():
                raise RuntimeError(f"list_consumed must be a contiguous 4-byte integer tensor of {tiles} tiles on {dev}")
            keep.append(list_consumed)
            a.list_consumed = list_consumed.data_ptr()
        if blend_pairs is not None:  # diagnostic: [T], (band, entry) pairs the blend evaluated (fovraster.h)
            tiles = ((W + 15) // 16) * ((H + 15) // 16)
            if blend_pairs.device != dev or blend_pairs.numel() != tiles or blend_pairs.element_size() != 4 or not blend_pairs.is_contiguous():
                raise RuntimeError(f"blend_pairs must be a contiguous 4-byte integer tensor of {tiles} tiles on {dev}")
            keep.append(blend_pairs)
            a.blend_pairs = blend_pairs.data_ptr()
        if packed is not None:
            packed.check(P, dev)
            if packed.produced is not None:
                packed.produced.wait_on_current(dev)
            a.packed_geom = _ptr(packed.geom)
            a.packed_colour = _ptr(packed.colour)
            a.packed_cull = _ptr(packed.cull)
            keep.append(packed)
        a.no_stats = int(bool(no_stats))
        # Inference frames keep to their own stream (fovraster.h): frames of successive calls share the GPU on several internal streams, a
        # process's streams share four hardware queues, and the library's helper streams -- even idle ones, even those of the caller's
        # own stream -- cost the overlapped frames 12 % (2170 -> 1870-1920 frames/s, tools/overlap9.py) for the 1 % they give a frame
        # that has the GPU to itself (fills and the short lists' sort beside the main stream: 1725 against 1718 frames/s).
        a.no_helper_streams = int(persistent and INFERENCE_NO_HELPER_STREAMS)
        a.emit_regions = int(bool(EMIT_REGIONS))
        if variant in (_native.VARIANT_PCHECK_OBB_SUM, _native.VARIANT_PCHECK_OBB_MAX, _native.VARIANT_PCHECK_OBB_LWMC) and not no_stats:
            counts = torch.empty((P,), dtype=torch.int32, device=dev)      # zeroed by fr_forward itself
            contribs = torch.empty((P,), dtype=torch.float32, device=dev)
            a.gaussians_count, a.contributions = counts.data_ptr(), contribs.data_ptr()
        a.geometry_resize, a.binning_resize, a.image_resize = ws.cbs[0], ws.cbs[1], ws.cbs[2]
        if _stage_events_hook is not None:
            a.stage_events = _stage_events_hook()
        handle = C.c_void_p()
        rc = lib.fr_forward_begin(C.byref(a), C.byref(handle))
        if rc != 0:
            raise RuntimeError(f"fovraster forward failed ({rc}): {_native.last_error()}")
    if reuse is not None:
        reuse.clear()
        if reuse_key is not None and counts is None and lease is None:  # (no per-call outputs beside the image and the radii)
            # (the caller's own tensors are NOT kept: the key vouches that the caller still holds the very objects the pointers
            # belong to; keeping them here would pin a dropped model's gigabytes until the next call)
            reuse.update(key=reuse_key, a=a, keep=own, ws=ws)
    return FrameInFlight(lib, a, keep, color, radii, ws, lease, counts, contribs, handle, dev, stream)


# ---- successive inference frames overlap on the GPU -------------------------------------------------------------------------------
# A frame is seven dependent stages bound by seven different units (HBM streaming, latency, HBM random access, the address unit, LDS,
# VALU: DESIGN 4); inside one frame nothing overlaps, and a host that renders one frame per call leaves every unit idle most of the
# time. The call itself cannot return before its instance count is in (the reference's contract: num_rendered), but nothing says the
# GPU must have finished frame n's sort and blend before it starts frame n + 1's cull pass. So an INFERENCE call (no autograd graph,
# persistent workspaces) runs on one of OVERLAP_SLOTS (3) internal streams in turn, each with its own workspace set: the head of call
# n + 1 goes to a stream whose last frame (n - 2) is long finished and runs beside the tail of call n (two streams: 1715 -> 1990
# frames/s on the S-6M bench frames; three, where the head never queues behind the tail of call n - 1: 2080). The caller's stream waits (on the GPU,
# not the host) for the frame's last kernel before anything enqueued after the call, so every use of the outputs is ordered as if
# the frame had run on the caller's stream.
# What makes this safe is the dependency on the INPUTS: a frame may only skip waiting for the caller's stream if nothing the caller
# enqueued since the previous call can have written what it reads. That is decided per call from the inputs' identity: the same tensor
# objects, storage addresses and autograd version counters as at the previous call of this (device, stream, thread) => unchanged, no
# wait; anything else (a model whose getters build new tensors per call, an optimiser step, another camera object) => the internal
# stream first waits for an event recorded on the caller's stream now -- the frame is then serialised behind whatever the caller
# enqueued, exactly as without this scheme. Writes the version counter cannot see (`.data` writes, raw-pointer kernels, DLPack /
# numpy aliases) are the caller's to announce: invalidate_overlap() (or OVERLAP_SUCCESSIVE_FRAMES = False).
# Images, radii and lists are bit-identical with and without (tested); only WHEN the kernels run changes.
OVERLAP_SUCCESSIVE_FRAMES = os.environ.get("FOVRASTER_OVERLAP", "1") != "0"  # (FOVRASTER_OVERLAP=0: developer A / B runs of single kernels)
_overlap_state = {}
_overlap_tls = threading.local()


class serial_frames:
    """with serial_frames(): ... -- calls inside run on the caller's stream alone (a caller that brackets the call with its own
    events -- render(starter=, ender=) -- measures the call's own kernels, not a neighbour's)."""

    def __enter__(self):
        _overlap_tls.off = getattr(_overlap_tls, "off", 0) + 1

    def __exit__(self, *exc):
        _overlap_tls.off -= 1


def invalidate_overlap():
    """The next inference call of every stream waits for its caller's stream (after a write to an input that bypassed the
    autograd version counter)."""
    for st in _overlap_state.values():
        st.sig = None


# Experimental (fovraster.h: fr_forward_args.emit_regions): region-major emission -- same lists, 1.5 x instead of 3.7 x the payload written,
# slower at present. Off by default.
EMIT_REGIONS = False
INFERENCE_NO_HELPER_STREAMS = os.environ.get("FOVRASTER_INFERENCE_HELPERS", "0") != "1"  # (=1: developer A / B runs)
OVERLAP_SLOTS = max(2, int(os.environ.get("FOVRASTER_OVERLAP_SLOTS", "3")))  # internal streams (and workspace sets) the frames take turns on


class _OverlapState:
    def __init__(self, dev, caller):
        n = OVERLAP_SLOTS
        self.streams = [torch.cuda.Stream(dev) for _ in range(n)]
        self.done = [None] * n       # event behind the last frame of each internal stream
        self.must_wait = [None] * n  # event on the caller's stream each internal stream still has to wait for
        self.events = [torch.cuda.Event() for _ in range(n)]  # (re-recorded every turn: the caller's stream waited for the previous record when it was made)
        self.reuse = [dict() for _ in range(n)]  # per stream: the argument struct of its previous frame (_forward_begin)
        self.turn = 0
        self.sig = None
        self.refs = None


def _input_signature(tensors):
    return tuple((id(t), t._version, t.data_ptr(), tuple(t.shape), t.stride()) for t in tensors)


def _packed_key(pk):
    """what a PackedModel is to a reused argument struct: its three buffers' addresses (an id() could be another object's by now)"""
    return None if pk is None else tuple(None if t is None else t.data_ptr() for t in (pk.geom, pk.colour, pk.cull))


def _forward_overlapped(args, kw):
    """_forward_begin + finish of an inference call on the next internal stream (see above). -> the result tuple of _forward_native."""
    rs, tensors = args[1], [t for t in args if isinstance(t, torch.Tensor) and t.numel() > 0]
    tensors += [t for t in (rs.bg, rs.viewmatrix, rs.projmatrix, rs.campos, kw.get("loss_map"), kw.get("sh_rest")) if isinstance(t, torch.Tensor)]
    dev = args[2].device
    cur = torch.cuda.current_stream(dev)
    key = (dev, cur.cuda_stream, threading.get_ident())
    st = _overlap_state.get(key)
    if st is None:
        if len(_overlap_state) > 16:
            _overlap_state.clear()
        st = _overlap_state[key] = _OverlapState(dev, cur)
    sig = _input_signature(tensors)
    # (a caller-supplied OUTPUT -- the diagnostics list_consumed / blend_pairs -- may still be read by something the caller enqueued
    # after the previous call: such a frame always waits for the caller's stream)
    outputs = kw.get("list_consumed") is not None or kw.get("blend_pairs") is not None
    same = (not outputs and st.sig == sig and st.refs is not None and len(st.refs) == len(tensors)
            and all(r() is t for r, t in zip(st.refs, tensors)))
    if not same:
        # new or modified inputs: both internal streams wait for the caller's stream as it stands now before they read them
        ev = torch.cuda.Event()
        ev.record(cur)
        st.must_wait = [ev] * len(st.streams)
        st.sig = None if outputs else sig
        import weakref
        st.refs = [weakref.ref(t) for t in tensors]
    i = st.turn
    st.turn = (i + 1) % len(st.streams)
    own = st.streams[i]
    if st.must_wait[i] is not None:
        own.wait_event(st.must_wait[i])
        st.must_wait[i] = None
    # (torch.cuda.stream()'s context manager costs 10 us of host time a call, and the host is on this scheme's critical path: the next
    # frame's head is enqueued only after this call has returned)
    done = st.events[i]
    # everything else a call is made of: with the same tensors AND the same scalars as the previous call on this internal stream the
    # argument struct is used again
    rkey = None if outputs else (sig, args[0], int(rs.image_height), int(rs.image_width), float(rs.tanfovx), float(rs.tanfovy), float(rs.scale_modifier),
                                 int(rs.sh_degree), bool(rs.prefiltered), tuple(sorted((k, v) for k, v in kw.items() if isinstance(v, (int, float, bool)))),
                                 args[-1] if isinstance(args[-1], float) else None, _packed_key(kw.get("packed")), bool(EMIT_REGIONS))
    if torch.cuda.current_device() == dev.index:
        torch.cuda.set_stream(own)
        try:
            res = _forward_begin(*args, on_stream=own, reuse=st.reuse[i], reuse_key=rkey, **kw).finish(on_stream=True)
            done.record(own)
        finally:
            torch.cuda.set_stream(cur)
    else:  # (tensors on another device than the current one: the context managers restore both)
        with torch.cuda.device(dev), torch.cuda.stream(own):
            res = _forward_begin(*args, **kw).finish()
            done.record(own)
    st.done[i] = done
    cur.wait_event(done)  # everything the caller enqueues from here on sees the finished frame
    for t in res:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            t.record_stream(cur)
    return res


def _forward_native(*args, **kw):
    """-> (num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer[, gaussians_count, contributions], lease):
    both halves of the forward call back to back (arguments: _forward_begin). Inference calls (persistent workspaces, no debug
    mode, no stream capture) of successive frames overlap on the GPU (OVERLAP_SUCCESSIVE_FRAMES, see above)."""
    if (OVERLAP_SUCCESSIVE_FRAMES and kw.get("persistent") and not getattr(_overlap_tls, "off", 0) and not args[1].debug
            and args[2].is_cuda and not torch.cuda.is_current_stream_capturing()):
        return _forward_overlapped(args, kw)
    return _forward_begin(*args, **kw).finish()


# Test switch: the gradient tensors handed to fr_backward hold NaN instead of whatever the allocator had (an element the library fails
# to write -- it writes every one, zeros included -- then shows in any comparison). The GPU test session sets it (tests/conftest.py).
POISON_GRADIENTS = False


def _alloc_gradients(z, P, M0, Mrest, has_cov, has_col):
    """The dense gradient tensors of one backward call, in _backward_native's order; z(*shape) allocates."""
    return (z(P, 3), z(P, 3), z(P, 1), z(P, 6) if has_cov else None, z(P, 3) if has_col else None, z(P, M0, 3), z(P, 3), z(P, 4),
            z(P, Mrest, 3) if Mrest is not None else None)


# Opt-in: the gradient tensors of a training step allocated and CLEARED at the end of the forward call, on a side stream that waits for
# the forward's kernels (fr_backward_prefill, fr_backward_args.outputs_zeroed), so that the 1.5 GB of fills of a 6 M-Gaussian model do
# not run beside k_render_bwd (which pays 0.11 ms for the company). Measured (tools/train_ab.py, S-6M / S-6M-T): the backward pass gets
# 0.21 / 0.06 ms shorter, the image loss between the two calls 0.27 ms LONGER (its kernels read two images; a 1.5 GB stream of writes
# raises the latency of every read on the chip): 2.43 -> 2.50 / 4.37 -> 4.58 ms per step. k_render_bwd -- arithmetic -- stays the one
# kernel of the step a fill hides behind; the switch is for hosts with something arithmetic-bound between their two calls.
PREZERO_GRADIENTS = False
_side_streams = {}


def _side_stream(dev):
    st = _side_streams.get(dev)
    if st is None:
        st = _side_streams[dev] = torch.cuda.Stream(dev)
    return st


def prefill_gradients(dev, P, M0, Mrest, has_cov, has_col, has_sh, radii=None):
    """-> (tensors in _alloc_gradients' order, event): allocated on the current stream, zero-filled by ONE kernel on the side stream
    behind everything the current stream holds now; the event marks the end of the fill. radii: handed on as a C host filling in the
    whole of fr_backward_args would (fr_backward_prefill clears every tensor in full whatever the struct holds)."""
    lib = _native.load()
    with torch.cuda.device(dev):
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev)
        z = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        if POISON_GRADIENTS:
            z = lambda *shape: torch.full(shape, float("nan"), dtype=torch.float32, device=dev)
        g = _alloc_gradients(z, P, M0, Mrest, has_cov, has_col)
        a = _native.BackwardArgs()
        a.P, a.M = P, M0 + (Mrest or 0)
        a.radii = radii.data_ptr() if radii is not None else None
        # (fr_backward_prefill only looks at which of shs / shs_rest / colors_precomp are given, not at what they hold)
        a.shs = g[5].data_ptr() if has_sh else None
        a.shs_rest = g[8].data_ptr() if Mrest is not None else None
        a.colors_precomp = None
        a.dL_dmean3D, a.dL_dmean2D, a.dL_dopacity = g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr()
        a.dL_dcov3D = g[3].data_ptr() if g[3] is not None else None
        a.dL_dcolor = g[4].data_ptr() if g[4] is not None else None
        a.dL_dsh = g[5].data_ptr() if (has_sh and M0) else None
        a.dL_dscale, a.dL_drot = g[6].data_ptr(), g[7].data_ptr()
        a.dL_dsh_rest = g[8].data_ptr() if g[8] is not None else None
        side.wait_stream(main)
        evs = _call_events
        if evs is not None:  # profiling.NativeCallTimer: the fill's own duration, on the stream it runs on
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
        rc = lib.fr_backward_prefill(C.byref(a), side.cuda_stream)
        if rc != 0:
            raise RuntimeError(f"fovraster backward_prefill failed ({rc}): {_native.last_error()}")
        if evs is not None:
            e1.record(side)
            evs.append(("fill", e0, e1))
        for t in g:
            if t is not None:
                t.record_stream(side)  # (freed before the fill has run -- a graph dropped without backward --: not reused under it)
        ev = torch.cuda.Event()
        ev.record(side)
    return g, ev


# Extension (multi-GPU training): a callable(k, row_lo, row_hi, grads) that fr_backward calls on the host behind every one of
# GRADIENT_RANGES pieces of its per-Gaussian pass -- `grads` = {"means3D", "opacities", "scales", "rotations", "sh", "sh_rest"} -> the dense
# gradient tensors of the call (None where there is none), rows [row_lo, row_hi) of each complete on the current stream once it gets
# there. Installed by multiview.OverlappedGradientExchange; None = one piece, no calls.
GRADIENT_RANGE_HOOK = None
GRADIENT_RANGES = 4


def _backward_native(variant, rs, means3D, radii, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                     grad_out_color, sh, geomBuffer, num_rendered, binningBuffer, imgBuffer, sh_rest=None,
                     want_cov3D_grad=False, want_color_grad=False, raw_activations=False, row_sparse=False, num_candidates=0, blend_pairs=None,
                     prezeroed=None):
    """-> (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations[, dL_dsh_rest])
    (dL_dsh_rest only with split SH storage: then dL_dsh is the DC part [P,1,3]; dL_dcov3D / dL_dcolors are None unless
    cov3Ds_precomp / colors_precomp are given or want_cov3D_grad / want_color_grad)
    row_sparse (extension, fovraster.h): the tensors are COMPACT, [num_candidates, ...], row i = the gradient of the Gaussian
    vis_list[i] (visible_rows(): the forward call's list of cull survivors, increasing indices); nothing is zero-filled."""
    lib = _native.load()
    dev = means3D.device
    P = means3D.size(0)
    Pfull = P
    if row_sparse:
        P = int(num_candidates)  # rows of the gradient tensors
    H, W = grad_out_color.size(1), grad_out_color.size(2)
    a = _native.BackwardArgs()
    keep = []

    def put(name, t, small=False):
        t = _f32_small(t, dev) if small else _f32(t, dev)
        keep.append(t)
        setattr(a, name, _ptr(t))
        return t

    with torch.cuda.device(dev):
        sh_c = put("shs", sh)
        rest_c = put("shs_rest", sh_rest)
        M0 = 0 if sh_c is None else sh_c.size(1)
        M = M0 + (0 if rest_c is None else rest_c.size(1))
        # every gradient tensor is written in full by fr_backward (zero rows included): no zero fill here (the reference
        # zero-fills 1.8 GB per step at 6 M Gaussians); P == 0 never reaches the library, hence zeros for that case
        z = (lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)) if P != 0 else \
            (lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev))
        if POISON_GRADIENTS and P != 0:
            z = lambda *shape: torch.full(shape, float("nan"), dtype=torch.float32, device=dev)
        # outputs only when the 3D covariances / colours are inputs; otherwise intermediates the library keeps per
        # visible Gaussian in its geometry workspace
        has_cov = want_cov3D_grad or (cov3Ds_precomp is not None and cov3Ds_precomp.numel() != 0)
        has_col = want_color_grad or (colors_precomp is not None and colors_precomp.numel() != 0)
        if prezeroed is not None:
            # allocated and zero-filled at the end of the forward call (prefill_gradients), beside the work between the two calls
            (dL_dmeans3D, dL_dmeans2D, dL_dopacity, dL_dcov3D, dL_dcolors, dL_dsh, dL_dscales, dL_drotations, dL_dsh_rest) = prezeroed
        else:
            (dL_dmeans3D, dL_dmeans2D, dL_dopacity, dL_dcov3D, dL_dcolors, dL_dsh, dL_dscales, dL_drotations, dL_dsh_rest) = \
                _alloc_gradients(z, P, M0, M - M0 if rest_c is not None else None, has_cov, has_col)
        if P != 0:
            a.variant, a.P, a.D, a.M, a.R = variant, Pfull, int(rs.sh_degree), M, int(num_rendered)
            a.W, a.H, a.debug = W, H, int(bool(rs.debug))
            a.raw_activations = int(bool(raw_activations))
            a.row_sparse = int(bool(row_sparse))
            a.outputs_zeroed = int(prezeroed is not None)
            a.tanfovx, a.tanfovy, a.scale_modifier = float(rs.tanfovx), float(rs.tanfovy), float(rs.scale_modifier)
            a.stream = torch.cuda.current_stream(dev).cuda_stream
            put("background", rs.bg, small=True)
            put("means3D", means3D)
            put("colors_precomp", colors_precomp)
            put("opacities", opacities)
            put("scales", scales)
            put("rotations", rotations)
            put("cov3D_precomp", cov3Ds_precomp)
            put("viewmatrix", rs.viewmatrix, small=True)
            put("projmatrix", rs.projmatrix, small=True)
            put("campos", rs.campos, small=True)
            put("dL_dpix", grad_out_color)
            a.radii = radii.data_ptr()
            a.geometry, a.binning, a.image = geomBuffer.data_ptr(), _ptr(binningBuffer if binningBuffer.numel() else None), imgBuffer.data_ptr()
            a.dL_dmean2D, a.dL_dconic, a.dL_dopacity = dL_dmeans2D.data_ptr(), None, dL_dopacity.data_ptr()
            a.dL_dcolor, a.dL_dmean3D = (dL_dcolors.data_ptr() if dL_dcolors is not None else None), dL_dmeans3D.data_ptr()
            a.dL_dcov3D = dL_dcov3D.data_ptr() if dL_dcov3D is not None else None
            a.dL_dsh = dL_dsh.data_ptr() if M else None
            a.dL_dsh_rest = dL_dsh_rest.data_ptr() if dL_dsh_rest is not None else None
            a.dL_dscale, a.dL_drot = dL_dscales.data_ptr(), dL_drotations.data_ptr()
            if _bwd_events_hook is not None:
                a.stage_events = _bwd_events_hook()
            if blend_pairs is not None:
                keep.append(blend_pairs)
                a.blend_pairs = blend_pairs.data_ptr()
            hook, hook_errors = GRADIENT_RANGE_HOOK, []
            if hook is not None and not row_sparse:
                # the per-Gaussian half of the call in pieces over ranges of rows; behind each piece the hook is told that those rows of
                # every gradient tensor are complete on this stream (fovraster.h: fr_backward_args.range_done) -- multiview's
                # OverlappedGradientExchange starts their all-reduce on its communication stream while the later pieces run
                grads = {"means3D": dL_dmeans3D, "opacities": dL_dopacity, "scales": dL_dscales, "rotations": dL_drotations,
                         "sh": dL_dsh if M else None, "sh_rest": dL_dsh_rest}

                def _range_done(_user, k, lo, hi):
                    try:
                        hook(int(k), int(lo), int(hi), grads)
                    except Exception as ex:  # (never let an exception cross the C frame)
                        hook_errors.append(ex)
                cb = _native.RANGE_FN(_range_done)
                keep.append(cb)
                a.num_ranges = int(GRADIENT_RANGES)
                a.range_done = C.cast(cb, C.c_void_p)
            rc = lib.fr_backward(C.byref(a))
            if rc != 0:
                raise RuntimeError(f"fovraster backward failed ({rc}): {_native.last_error()}")
            if hook_errors:
                raise hook_errors[0]
    out = (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)
    return out + (dL_dsh_rest,) if dL_dsh_rest is not None else out


def visible_rows(variant, P, geomBuffer, num_candidates):
    """The Gaussian indices (int64 [num_candidates], increasing) of the rows of a row-sparse backward call."""
    lib = _native.load()
    off = lib.fr_geometry_vis_list(variant, P, geomBuffer.data_ptr()) - geomBuffer.data_ptr()
    return geomBuffer[off:off + 4 * int(num_candidates)].view(torch.int32).long()


def _mark_visible(positions, rs):
    lib = _native.load()
    _require_gpu(positions)
    dev = positions.device
    P = positions.size(0)
    present = torch.zeros((P,), dtype=torch.bool, device=dev)
    if P:
        pos, vm, pm = _f32(positions, dev), _f32(rs.viewmatrix, dev), _f32(rs.projmatrix, dev)
        with torch.cuda.device(dev):
            rc = lib.fr_mark_visible(P, pos.data_ptr(), vm.data_ptr(), pm.data_ptr(), present.data_ptr(),
                                     torch.cuda.current_stream(dev).cuda_stream)
        if rc != 0:
            raise RuntimeError(f"fovraster mark_visible failed ({rc}): {_native.last_error()}")
    return present


def _make_plain(variant_id, with_counts, has_backward, takes_loss_map=False):
    """Autograd function + module for the non-foveated variants. takes_loss_map: the
    …_loss_weighted_max_count extension has one extra input (`loss_map`, a [3,H,W] or [H,W] tensor)."""

    class _RasterizeGaussians(torch.autograd.Function):
        @staticmethod
        def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                    raster_settings, loss_map=None, sh_rest=None, packed=None, grad_mode=True, raw_activations=False,
                    row_sparse=False, want_stats=True):
            # want_stats=False (extension, pcheck_obb_sum): the caller drops gaussians_count / contributions (eff_finetune.py:107-108
            # does): the blend skips them and the call returns (color, radii) only
            # row_sparse (extension): backward returns the gradients of the [P, ...] inputs as SPARSE tensors (torch.sparse_coo, one
            # sparse dimension: the Gaussians this view touched) -- for leaf parameters (raw_activations + split SH: every input is
            # one) optimised by a sparse-aware optimizer or summed with multiview.allreduce_gradients; no 1.5 GB of zero fills
            # raw_activations (extension): opacities / scales / rotations are the model's RAW parameters, the kernels apply
            # sigmoid / exp / normalize themselves and the backward pass returns the gradients w.r.t. the raw parameters
            # sh_rest (extension): the SH coefficients as the two tensors a model stores, sh = features_dc
            # [P,1,3], sh_rest = features_rest [P,M-1,3]; saves the torch.cat of get_features and its backward
            args = (variant_id, raster_settings, means3D, sh, colors_precomp, opacities, scales, rotations,
                    cov3Ds_precomp)
            if sh_rest is not None and sh_rest.numel() == 0:
                sh_rest = None
            # backward re-reads the workspaces; needs_input_grad reflects requires_grad even under torch.no_grad(), where
            # no graph is built: evaluation renders of a model with Parameters use the persistent inference set
            # (grad_mode = torch.is_grad_enabled() at the call site: inside forward() it is always off)
            keep_ws = has_backward and grad_mode and any(ctx.needs_input_grad)
            if takes_loss_map:
                if loss_map is None or loss_map.numel() < raster_settings.image_height * raster_settings.image_width:
                    raise Exception("loss_map with at least image_height*image_width values is required")
            no_stats = with_counts and not want_stats and variant_id == _native.VARIANT_PCHECK_OBB_SUM
            ctx.split_sh = sh_rest is not None
            ctx.raw_activations = bool(raw_activations)
            ctx.row_sparse = bool(row_sparse) and has_backward
            # no zero tensors for the gradients of the outputs nobody differentiates (radii, counts, contributions:
            # three [P] fills per step otherwise)
            ctx.set_materialize_grads(False)
            def run():
                ev = _call_events
                if ev is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                frame = _forward_begin(*args, persistent=not keep_ws, loss_map=loss_map, sh_rest=sh_rest, packed=packed,
                                       raw_activations=raw_activations, no_stats=no_stats)
                out = frame.finish()
                if ev is not None:
                    e1.record()
                    ev.append(("fwd", e0, e1))
                ctx.num_candidates = int(frame.a.num_candidates)
                return out
            if raster_settings.debug:
                cpu_args = cpu_deep_copy_tuple(args)  # copy them before they can be corrupted
                try:
                    res = run()
                except Exception as ex:
                    torch.save(cpu_args, "snapshot_fw.dump")
                    print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                    raise ex
            else:
                res = run()
            num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = res[:6]
            if not keep_ws:  # nothing will call backward: do not pin the shared workspaces
                geomBuffer = binningBuffer = imgBuffer = torch.empty(0, dtype=torch.uint8, device=means3D.device)
            ctx.raster_settings = raster_settings
            ctx.num_rendered = num_rendered
            ctx.ws_lease = res[-1] if keep_ws else None
            ctx.prezero = None
            if keep_ws and PREZERO_GRADIENTS and not ctx.row_sparse and means3D.size(0) > 0:
                has_sh = sh.numel() != 0
                M0 = sh.size(1) if has_sh else 0
                ctx.prezero = prefill_gradients(means3D.device, means3D.size(0), M0, sh_rest.size(1) if sh_rest is not None else None,
                                                cov3Ds_precomp.numel() != 0, colors_precomp.numel() != 0, has_sh, radii=radii)
            ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, opacities,
                                  geomBuffer, binningBuffer, imgBuffer,
                                  sh_rest if sh_rest is not None else torch.empty(0, device=means3D.device))
            ctx.mark_non_differentiable(radii)
            if with_counts and not no_stats:
                ctx.mark_non_differentiable(res[6], res[7])
                return color, radii, res[6], res[7]
            return color, radii

        @staticmethod
        def backward(ctx, grad_out_color, *_unused):
            if not has_backward:
                # the reference's inference-only extension exports no backward entry point
                raise RuntimeError("this rasterizer variant is inference-only (no backward in the reference)")
            if grad_out_color is None:  # the image took no part in the loss
                return (None,) * 16
            rs = ctx.raster_settings
            (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, opacities,
             geomBuffer, binningBuffer, imgBuffer, sh_rest) = ctx.saved_tensors
            args = (variant_id, rs, means3D, radii, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                    grad_out_color, sh, geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer,
                    sh_rest if ctx.split_sh else None)
            kw = dict(raw_activations=ctx.raw_activations, row_sparse=ctx.row_sparse, num_candidates=ctx.num_candidates)
            pre, ctx.prezero = ctx.prezero, None  # (a second backward over the same graph allocates its own tensors)
            if pre is not None:
                torch.cuda.current_stream(means3D.device).wait_event(pre[1])
                kw["prezeroed"] = pre[0]
            ev = _call_events
            if ev is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if rs.debug:
                cpu_args = cpu_deep_copy_tuple(args)
                try:
                    res = _backward_native(*args, **kw)
                except Exception as ex:
                    torch.save(cpu_args, "snapshot_bw.dump")
                    print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                    raise ex
            else:
                res = _backward_native(*args, **kw)
            if ev is not None:
                e1.record()
                ev.append(("bwd", e0, e1))
            if ctx.row_sparse:
                # compact rows -> sparse tensors of the inputs' shapes (indices shared: one [1, C] tensor)
                P = means3D.size(0)
                rows = visible_rows(variant_id, P, geomBuffer, ctx.num_candidates).unsqueeze(0)
                res = tuple(None if g is None else torch.sparse_coo_tensor(rows, g, (P,) + tuple(g.shape[1:]), is_coalesced=True)
                            for g in res)
            (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh,
             grad_scales, grad_rotations) = res[:8]
            grads = (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales,
                     grad_rotations, grad_cov3Ds_precomp, None)
            # loss_map, sh_rest, packed, grad_mode, raw_activations, row_sparse, want_stats
            return grads + (None, res[8] if ctx.split_sh else None, None, None, None, None, None)

    def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                            raster_settings, loss_map=None, sh_rest=None, packed=None, raw_activations=False, row_sparse=False,
                            want_stats=True):
        return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                         cov3Ds_precomp, raster_settings, loss_map if takes_loss_map else None, sh_rest, packed,
                                         torch.is_grad_enabled(), raw_activations, row_sparse, want_stats)

    class GaussianRasterizer(nn.Module):
        def __init__(self, raster_settings):
            super().__init__()
            self.raster_settings = raster_settings

        def markVisible(self, positions):
            with torch.no_grad():
                return _mark_visible(positions, self.raster_settings)

        def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                    cov3D_precomp=None, loss_map=None, packed=None, raw_activations=False, row_sparse=False, want_stats=True):
            raster_settings = self.raster_settings
            if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
                raise Exception('Please provide excatly one of either SHs or precomputed colors!')
            if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                    ((scales is not None or rotations is not None) and cov3D_precomp is not None):
                raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
            if raw_activations and (cov3D_precomp is not None or packed is not None):
                raise Exception('raw_activations needs the scale/rotation pair and no packed model')
            empty = torch.Tensor([])
            shs_rest = None
            if isinstance(shs, (tuple, list)):  # extension: (features_dc [P,1,3], features_rest [P,M-1,3])
                shs, shs_rest = shs
            shs = empty if shs is None else shs
            colors_precomp = empty if colors_precomp is None else colors_precomp
            scales = empty if scales is None else scales
            rotations = empty if rotations is None else rotations
            cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
            return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                       cov3D_precomp, raster_settings, loss_map, shs_rest, packed, raw_activations, row_sparse, want_stats)

    return _RasterizeGaussians, rasterize_gaussians, GaussianRasterizer


def _gaze_pair(gazeArray):
    if gazeArray is None:
        raise Exception("gazeArray is required by the foveated rasterizer")
    if isinstance(gazeArray, torch.Tensor):
        g = gazeArray.detach().flatten().tolist()  # the reference does two .item() syncs here
    else:
        g = list(gazeArray)
    return float(g[0]), float(g[1])


def _make_fov():
    """Autograd function + module for the foveated (inference-only) variant."""
    variant_id = _native.VARIANT_FOV_PCHECK_OBB

    class _RasterizeGaussians(torch.autograd.Function):
        @staticmethod
        def forward(ctx, means3D, means2D, shs_rest, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                    raster_settings, shs_dcs, highest_levels, gazeArray, alpha, blending, packed=None):
            args = (variant_id, raster_settings, means3D, shs_rest, colors_precomp, opacities, scales, rotations,
                    cov3Ds_precomp, shs_dcs, highest_levels, _gaze_pair(gazeArray), float(alpha))
            if raster_settings.debug:
                cpu_args = cpu_deep_copy_tuple(args)
                try:
                    res = _forward_native(*args, persistent=True, packed=packed)
                except Exception as ex:
                    torch.save(cpu_args, "snapshot_fw.dump")
                    print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                    raise ex
            else:
                res = _forward_native(*args, persistent=True, packed=packed)
            num_rendered, color, radii = res[:3]
            ctx.num_rendered = num_rendered
            ctx.mark_non_differentiable(radii)
            return color, radii

        @staticmethod
        def backward(ctx, grad_out_color, _):
            # RF/diff_gaussian_rasterization_fov_pcheck_obb/__init__.py:128-187: the foveated extension is
            # inference-only and hands back None for every input
            return (None,) * 15

    def rasterize_gaussians(means3D, means2D, shs_rest, colors_precomp, opacities, scales, rotations,
                            cov3Ds_precomp, raster_settings, shs_dcs, highest_levels, gazeArray, alpha, blending,
                            packed=None):
        if not torch.is_grad_enabled() and not raster_settings.debug:
            # (inference: no graph to build -- the autograd machinery around an inference-only extension is 15 us of host time a
            # frame on a path where the host sets the pace, see OVERLAP_SUCCESSIVE_FRAMES)
            res = _forward_native(variant_id, raster_settings, means3D, shs_rest, colors_precomp, opacities, scales, rotations,
                                  cov3Ds_precomp, shs_dcs, highest_levels, _gaze_pair(gazeArray), float(alpha), persistent=True, packed=packed)
            return res[1], res[2]
        return _RasterizeGaussians.apply(means3D, means2D, shs_rest, colors_precomp, opacities, scales, rotations,
                                         cov3Ds_precomp, raster_settings, shs_dcs, highest_levels, gazeArray, alpha,
                                         blending, packed)

    class GaussianRasterizer(nn.Module):
        def __init__(self, raster_settings):
            super().__init__()
            self.raster_settings = raster_settings

        def markVisible(self, positions):
            with torch.no_grad():
                return _mark_visible(positions, self.raster_settings)

        def forward(self, means3D, means2D, opacities, shs_rest=None, colors_precomp=None, scales=None,
                    rotations=None, cov3D_precomp=None, shs_dcs=None, highest_levels=None, gazeArray=None,
                    alpha=None, blending=None, packed=None):
            raster_settings = self.raster_settings
            if (shs_rest is None and colors_precomp is None) or (shs_rest is not None and colors_precomp is not None):
                raise Exception('Please provide excatly one of either SHs or precomputed colors!')
            if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                    ((scales is not None or rotations is not None) and cov3D_precomp is not None):
                raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
            empty = torch.Tensor([])
            shs_rest = empty if shs_rest is None else shs_rest
            colors_precomp = empty if colors_precomp is None else colors_precomp
            scales = empty if scales is None else scales
            rotations = empty if rotations is None else rotations
            cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
            return rasterize_gaussians(means3D, means2D, shs_rest, colors_precomp, opacities, scales, rotations,
                                       cov3D_precomp, raster_settings, shs_dcs, highest_levels, gazeArray, alpha,
                                       blending, packed)

    return _RasterizeGaussians, rasterize_gaussians, GaussianRasterizer


def _make_naive_fov():
    """Autograd function + module for the shared-model foveated baseline ("SMFR"; reference package
    …_naive_pcheck_obb/diff_gaussian_rasterization_naive_pcheck_obb/__init__.py:20-262): the foveated extension's
    tile levels and lists with the plain model's single colour / opacity per Gaussian. Inference only."""
    variant_id = _native.VARIANT_NAIVE_FOV_PCHECK_OBB

    class _RasterizeGaussians(torch.autograd.Function):
        @staticmethod
        def forward(ctx, means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                    raster_settings, highest_levels, gazeArray, alpha, blending):
            args = (variant_id, raster_settings, means3D, shs, colors_precomp, opacities, scales, rotations,
                    cov3Ds_precomp, None, highest_levels, _gaze_pair(gazeArray), float(alpha))
            if raster_settings.debug:
                cpu_args = cpu_deep_copy_tuple(args)
                try:
                    res = _forward_native(*args, persistent=True)
                except Exception as ex:
                    torch.save(cpu_args, "snapshot_fw.dump")
                    print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                    raise ex
            else:
                res = _forward_native(*args, persistent=True)
            num_rendered, color, radii = res[:3]
            ctx.num_rendered = num_rendered
            ctx.mark_non_differentiable(radii)
            return color, radii

        @staticmethod
        def backward(ctx, grad_out_color, _):
            # the reference's backward() of this extension returns None for every input (its kernels are forward-only)
            return (None,) * 13

    def rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                            raster_settings, highest_levels, gazeArray, alpha, blending):
        return _RasterizeGaussians.apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                         cov3Ds_precomp, raster_settings, highest_levels, gazeArray, alpha, blending)

    class GaussianRasterizer(nn.Module):
        def __init__(self, raster_settings):
            super().__init__()
            self.raster_settings = raster_settings

        def markVisible(self, positions):
            with torch.no_grad():
                return _mark_visible(positions, self.raster_settings)

        def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                    cov3D_precomp=None, shs_dcs=None, highest_levels=None, gazeArray=None, alpha=None, blending=None):
            raster_settings = self.raster_settings
            if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
                raise Exception('Please provide excatly one of either SHs or precomputed colors!')
            if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                    ((scales is not None or rotations is not None) and cov3D_precomp is not None):
                raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
            empty = torch.Tensor([])
            shs = empty if shs is None else shs
            colors_precomp = empty if colors_precomp is None else colors_precomp
            scales = empty if scales is None else scales
            rotations = empty if rotations is None else rotations
            cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
            return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                       cov3D_precomp, raster_settings, highest_levels, gazeArray, alpha, blending)

    return _RasterizeGaussians, rasterize_gaussians, GaussianRasterizer


_level_zeros = {}


def _zero_levels(P, device):
    """[P,1] zeros: the multi-model variant's stand-in for highest_levels (its skip test reuses the level filter)."""
    key = (P, device)
    ent = _level_zeros.get(key)
    if ent is None:
        _level_zeros.clear()
        t = torch.zeros((P, 1), dtype=torch.float32, device=device)
        ent = _level_zeros[key] = (t, _Produced(device, (t,)))
    ent[1].wait_on_current(device)  # (the fill ran on the stream that was current when the buffer was made)
    return ent[0]


def _make_mmfr():
    """Autograd function + module for ONE level of the multi-model foveated baseline ("MMFR"; reference package
    …_mmfr_pcheck_obb/diff_gaussian_rasterization_mmfr_pcheck_obb/__init__.py:20-262). Inference only."""
    variant_id = _native.VARIANT_MMFR_PCHECK_OBB

    class _RasterizeGaussians(torch.autograd.Function):
        @staticmethod
        def forward(ctx, means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                    raster_settings, cur_level, gazeArray, alpha, blending):
            if not means3D.is_cuda:
                _require_gpu(means3D)
            args = (variant_id, raster_settings, means3D, shs, colors_precomp, opacities, scales, rotations,
                    cov3Ds_precomp, None, _zero_levels(means3D.size(0), means3D.device), _gaze_pair(gazeArray), float(alpha))
            if raster_settings.debug:
                cpu_args = cpu_deep_copy_tuple(args)
                try:
                    res = _forward_native(*args, persistent=True, cur_level=float(cur_level))
                except Exception as ex:
                    torch.save(cpu_args, "snapshot_fw.dump")
                    print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                    raise ex
            else:
                res = _forward_native(*args, persistent=True, cur_level=float(cur_level))
            num_rendered, color, radii = res[:3]
            ctx.num_rendered = num_rendered
            ctx.mark_non_differentiable(radii)
            return color, radii

        @staticmethod
        def backward(ctx, grad_out_color, _):
            return (None,) * 13

    def rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                            raster_settings, cur_level, gazeArray, alpha, blending):
        return _RasterizeGaussians.apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                         cov3Ds_precomp, raster_settings, cur_level, gazeArray, alpha, blending)

    class GaussianRasterizer(nn.Module):
        def __init__(self, raster_settings):
            super().__init__()
            self.raster_settings = raster_settings

        def markVisible(self, positions):
            with torch.no_grad():
                return _mark_visible(positions, self.raster_settings)

        def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                    cov3D_precomp=None, shs_dcs=None, cur_level=None, gazeArray=None, alpha=None, blending=None):
            raster_settings = self.raster_settings
            if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
                raise Exception('Please provide excatly one of either SHs or precomputed colors!')
            if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                    ((scales is not None or rotations is not None) and cov3D_precomp is not None):
                raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
            empty = torch.Tensor([])
            shs = empty if shs is None else shs
            colors_precomp = empty if colors_precomp is None else colors_precomp
            scales = empty if scales is None else scales
            rotations = empty if rotations is None else rotations
            cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
            return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                       cov3D_precomp, raster_settings, cur_level, gazeArray, alpha, blending)

    return _RasterizeGaussians, rasterize_gaussians, GaussianRasterizer
