"""ctypes binding of libfovraster_hip.so (C ABI declared in include/fovraster.h).

There is NO CPU fallback: if the HIP library is missing or cannot be loaded, every rasterizer
call raises. Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C fov-3dgs_amd/csrc``.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# (FOVRASTER_LIB: an experiment build of the library, tools/ab_build.sh -- never a different implementation: same ABI check)
LIB_PATH = os.environ.get("FOVRASTER_LIB") or os.path.join(HERE, "libfovraster_hip.so")
ABI_VERSION = 9

VARIANT_ORIGINAL, VARIANT_PCHECK_OBB_SUM, VARIANT_PCHECK_OBB, VARIANT_FOV_PCHECK_OBB = 0, 1, 2, 3
VARIANT_PCHECK_OBB_MAX, VARIANT_PCHECK_OBB_LWMC, VARIANT_NAIVE_FOV_PCHECK_OBB, VARIANT_MMFR_PCHECK_OBB = 4, 5, 6, 7
STAGES = ("tile_levels", "project", "bin", "tile_scan", "emit", "tile_sort", "render")
NUM_STAGE_EVENTS = len(STAGES) + 1
VARIANT_IDS = {"original": 0, "pcheck_obb_sum": 1, "pcheck_obb": 2, "fov_pcheck_obb": 3, "pcheck_obb_max": 4,
               "pcheck_obb_loss_weighted_max_count": 5, "naive_pcheck_obb": 6, "mmfr_pcheck_obb": 7}

RESIZE_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)

_FP = C.c_void_p  # device pointers are passed as raw addresses


class ForwardArgs(C.Structure):
    _fields_ = [
        ("variant", C.c_int32), ("P", C.c_int32), ("D", C.c_int32), ("M", C.c_int32),
        ("W", C.c_int32), ("H", C.c_int32), ("prefiltered", C.c_int32), ("debug", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
        ("gaze_x", C.c_float), ("gaze_y", C.c_float), ("alpha", C.c_float),
        ("stream", C.c_void_p),
        ("background", _FP), ("means3D", _FP), ("shs", _FP), ("colors_precomp", _FP), ("opacities", _FP),
        ("scales", _FP), ("rotations", _FP), ("cov3D_precomp", _FP), ("viewmatrix", _FP), ("projmatrix", _FP),
        ("campos", _FP), ("shs_dcs", _FP), ("highest_levels", _FP),
        ("out_color", _FP), ("radii", _FP), ("gaussians_count", _FP), ("contributions", _FP),
        ("geometry_resize", RESIZE_FN), ("binning_resize", RESIZE_FN), ("image_resize", RESIZE_FN),
        ("resize_user", C.c_void_p * 3),
        ("num_rendered", C.c_int32), ("max_tile_instances", C.c_int32),
        ("stage_events", C.POINTER(C.c_void_p)),
        ("loss_map", _FP),
        ("shs_rest", _FP),
        ("packed_geom", _FP), ("packed_colour", _FP), ("packed_cull", _FP),
        ("cur_level", C.c_float),
        ("raw_activations", C.c_int32),
        ("num_candidates", C.c_int32),
        ("list_consumed", _FP),
        ("no_stats", C.c_int32),
        ("blend_pairs", _FP),
        ("no_helper_streams", C.c_int32),
        ("emit_regions", C.c_int32),
    ]


class BackwardArgs(C.Structure):
    _fields_ = [
        ("variant", C.c_int32), ("P", C.c_int32), ("D", C.c_int32), ("M", C.c_int32), ("R", C.c_int32),
        ("W", C.c_int32), ("H", C.c_int32), ("debug", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
        ("stream", C.c_void_p),
        ("background", _FP), ("means3D", _FP), ("shs", _FP), ("colors_precomp", _FP), ("opacities", _FP),
        ("scales", _FP), ("rotations", _FP), ("cov3D_precomp", _FP), ("viewmatrix", _FP), ("projmatrix", _FP),
        ("campos", _FP),
        ("radii", _FP), ("geometry", _FP), ("binning", _FP), ("image", _FP), ("dL_dpix", _FP),
        ("dL_dmean2D", _FP), ("dL_dconic", _FP), ("dL_dopacity", _FP), ("dL_dcolor", _FP), ("dL_dmean3D", _FP),
        ("dL_dcov3D", _FP), ("dL_dsh", _FP), ("dL_dscale", _FP), ("dL_drot", _FP),
        ("stage_events", C.POINTER(C.c_void_p)),
        ("shs_rest", _FP),
        ("dL_dsh_rest", _FP),
        ("raw_activations", C.c_int32),
        ("row_sparse", C.c_int32),
        ("blend_pairs", _FP),
        ("outputs_zeroed", C.c_int32),
        ("num_ranges", C.c_int32),
        ("range_done", C.c_void_p),   # RANGE_FN, set through ctypes.cast (a NULL function pointer is the default)
        ("range_user", C.c_void_p),
    ]


RANGE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.c_int32, C.c_int32)  # fr_backward_args.range_done(user, k, row_lo, row_hi)


EXPORTS = ("fr_abi_version", "fr_last_error", "fr_event_create", "fr_event_destroy", "fr_event_elapsed_ms", "fr_forward", "fr_backward", "fr_mark_visible", "fr_pack_geom", "fr_pack_colour", "fr_pack_cull", "fr_activate_forward", "fr_activate_backward", "fr_l1_ssim_blocks", "fr_l1_ssim_forward", "fr_l1_ssim_finish", "fr_l1_ssim_backward",
           "fr_geometry_bytes", "fr_image_bytes", "fr_binning_bytes", "fr_image_ranges",
           "fr_binning_point_list", "fr_image_final_T", "fr_image_n_contrib", "fr_image_tile_levels", "fr_geometry_records",
           "fr_geometry_vis_list", "fr_geometry_vis_count", "fr_geometry_walk_records", "fr_geometry_level_colours",
           "fr_geometry_level_ranges", "fr_forward_begin", "fr_forward_finish", "fr_forward_abandon", "fr_backward_prefill")

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def load():
    """Load (once) and return the HIP library; raise NativeLibraryError if that is impossible."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"fovraster: {LIB_PATH} not found -- the HIP extension is not built and there is no CPU fallback. "
            "Run `make -C fov-3dgs_amd/csrc` (or __graft_entry__.build()).")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # missing ROCm runtime, wrong arch, ...
        raise NativeLibraryError(f"fovraster: cannot load {LIB_PATH}: {e}") from e
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise NativeLibraryError(f"fovraster: {LIB_PATH} does not export {name}")
    lib.fr_abi_version.restype = C.c_int
    lib.fr_last_error.restype = C.c_char_p
    lib.fr_event_create.restype = C.c_void_p
    lib.fr_event_destroy.argtypes = [C.c_void_p]
    lib.fr_event_destroy.restype = None
    lib.fr_event_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
    lib.fr_event_elapsed_ms.restype = C.c_int
    lib.fr_forward.argtypes = [C.POINTER(ForwardArgs)]
    lib.fr_forward.restype = C.c_int
    lib.fr_backward.argtypes = [C.POINTER(BackwardArgs)]
    lib.fr_backward.restype = C.c_int
    lib.fr_mark_visible.argtypes = [C.c_int32, _FP, _FP, _FP, _FP, C.c_void_p]
    lib.fr_mark_visible.restype = C.c_int
    lib.fr_pack_geom.argtypes = [C.c_int32, _FP, _FP, _FP, _FP, C.c_int32, _FP, _FP, C.c_void_p]
    lib.fr_pack_geom.restype = C.c_int
    lib.fr_pack_colour.argtypes = [C.c_int32, _FP, _FP, _FP, _FP, C.c_void_p]
    lib.fr_pack_colour.restype = C.c_int
    lib.fr_pack_cull.argtypes = [C.c_int32, _FP, _FP, _FP, _FP, C.c_void_p]
    lib.fr_pack_cull.restype = C.c_int
    lib.fr_activate_forward.argtypes = [C.c_int32] + [_FP] * 6 + [C.c_void_p]
    lib.fr_activate_forward.restype = C.c_int
    lib.fr_activate_backward.argtypes = [C.c_int32] + [_FP] * 9 + [C.c_void_p]
    lib.fr_activate_backward.restype = C.c_int
    lib.fr_l1_ssim_blocks.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.fr_l1_ssim_blocks.restype = C.c_int64
    lib.fr_l1_ssim_forward.argtypes = [C.c_int32, C.c_int32, C.c_int32, _FP, _FP, _FP, _FP, C.c_void_p]
    lib.fr_l1_ssim_forward.restype = C.c_int
    lib.fr_l1_ssim_finish.argtypes = [C.c_int32, C.c_int32, C.c_int32, _FP, C.c_float, _FP, C.c_void_p]
    lib.fr_l1_ssim_finish.restype = C.c_int
    lib.fr_l1_ssim_backward.argtypes = [C.c_int32, C.c_int32, C.c_int32, _FP, _FP, _FP, C.c_float, C.c_float, _FP, _FP, C.c_void_p]
    lib.fr_l1_ssim_backward.restype = C.c_int
    for n in ("fr_geometry_bytes",):
        getattr(lib, n).argtypes = [C.c_int32, C.c_int32]
        getattr(lib, n).restype = C.c_size_t
    lib.fr_image_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.fr_image_bytes.restype = C.c_size_t
    lib.fr_binning_bytes.argtypes = [C.c_int32, C.c_int64]
    lib.fr_binning_bytes.restype = C.c_size_t
    for n in ("fr_image_ranges", "fr_image_final_T", "fr_image_n_contrib"):
        getattr(lib, n).argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        getattr(lib, n).restype = C.c_void_p
    lib.fr_binning_point_list.argtypes = [C.c_int32, C.c_int64, C.c_void_p]
    lib.fr_binning_point_list.restype = C.c_void_p
    lib.fr_geometry_records.argtypes = [C.c_int32, C.c_int32, C.c_void_p]
    lib.fr_geometry_records.restype = C.c_void_p
    for n in ("fr_geometry_vis_list", "fr_geometry_vis_count", "fr_geometry_walk_records"):
        getattr(lib, n).argtypes = [C.c_int32, C.c_int32, C.c_void_p]
        getattr(lib, n).restype = C.c_void_p
    for n in ("fr_geometry_level_colours", "fr_geometry_level_ranges"):
        getattr(lib, n).argtypes = [C.c_int32, C.c_void_p]
        getattr(lib, n).restype = C.c_void_p
    lib.fr_image_tile_levels.argtypes = [C.c_int32, C.c_int32, C.c_void_p]
    lib.fr_image_tile_levels.restype = C.c_void_p
    lib.fr_forward_begin.argtypes = [C.POINTER(ForwardArgs), C.POINTER(C.c_void_p)]
    lib.fr_forward_begin.restype = C.c_int
    lib.fr_forward_finish.argtypes = [C.c_void_p]
    lib.fr_forward_finish.restype = C.c_int
    lib.fr_backward_prefill.argtypes = [C.POINTER(BackwardArgs), C.c_void_p]
    lib.fr_backward_prefill.restype = C.c_int
    lib.fr_forward_abandon.argtypes = [C.c_void_p]
    lib.fr_forward_abandon.restype = C.c_int
    if lib.fr_abi_version() != ABI_VERSION:
        raise NativeLibraryError(f"fovraster: ABI version mismatch ({lib.fr_abi_version()} != {ABI_VERSION})")
    _lib = lib
    return lib


def last_error():
    return load().fr_last_error().decode("utf-8", "replace")
