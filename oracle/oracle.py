"""ctypes/numpy front-end of the CPU oracle (oracle/fovraster_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.

`forward()` / `backward()` take plain dicts of numpy arrays that mirror the
arguments of the reference's `_C.rasterize_gaussians(_backward)`
and return dicts with every intermediate the reference keeps in its
geometry/binning/image state, so tests can compare stage by stage.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")
SRC = os.path.join(HERE, "fovraster_oracle.c")

VARIANTS = {"original": 0, "pcheck_obb_sum": 1, "pcheck_obb": 2, "fov_pcheck_obb": 3, "pcheck_obb_max": 4,
            "pcheck_obb_loss_weighted_max_count": 5, "naive_pcheck_obb": 6, "mmfr_pcheck_obb": 7}
FOV_NUM = 4


def _host_has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return " fma " in line + " "
    except OSError:
        pass
    return False


# The three flavours of the ONE source file:
#   f32      float, -ffp-contract=off: the literal reading of the reference's fp32 expressions (the checker of the parity tests);
#   f64      double (-DORC_DOUBLE): finite-difference pins, arithmetic-noise yardstick;
#   f32_fma  float, -ffp-contract=fast -mfma: every multiply whose result feeds an add / subtract is fused into it, within and
#            across statements -- what nvcc's default -fmad=true does to the reference's .cu files (its setup.py sets no
#            -fmad=false: R0/setup.py:12-29). gcc and nvcc need not pick the same product of `a*b + c*d`, so this flavour is not
#            "the reference binary"; it measures how far contraction CAN move radii, lists and pixels (tools/fma_envelope.py,
#            tests/test_second_derivation.py). Needs a host with FMA3 (the MI355X box's EPYC and this container have it).
FLAVOURS = {"f32": ("liboracle_f32.so", ["-ffp-contract=off"]), "f64": ("liboracle_f64.so", ["-ffp-contract=off", "-DORC_DOUBLE"]),
            "f32_fma": ("liboracle_f32_fma.so", ["-ffp-contract=fast", "-mfma"])}


def build(force=False):
    """Compile the C restatement (float, double and contracted-float flavours) with gcc."""
    os.makedirs(BUILD, exist_ok=True)
    outs = []
    for key, (name, flags) in FLAVOURS.items():
        if key == "f32_fma" and not _host_has_fma():
            continue
        out = os.path.join(BUILD, name)
        outs.append(out)
        if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(SRC):
            continue
        cmd = ["gcc", "-O2", "-fopenmp", "-fPIC", "-shared", "-Wall", *flags, "-o", out, SRC, "-lm"]
        subprocess.check_call(cmd)
    return outs


_LIBS = {}
_THREADS = 1


def set_threads(n):
    """Host threads the oracle's per-Gaussian / per-tile loops are spread over (default 1). Index outputs and images
    do not depend on it; per-Gaussian sums across tiles are accumulated in double by atomic adds."""
    global _THREADS
    _THREADS = max(1, int(n))
    for lib in _LIBS.values():
        lib.orc_set_threads(_THREADS)


def has_fma_flavour():
    return _host_has_fma()


def _lib(dtype, fma=False):
    size = np.dtype(dtype).itemsize
    key = "f64" if size == 8 else ("f32_fma" if fma else "f32")
    if fma and size != 4:
        raise ValueError("the contracted flavour exists in float only")
    if key not in _LIBS:
        if key == "f32_fma" and not _host_has_fma():
            raise RuntimeError("oracle: the contracted flavour needs a host with FMA3")
        build()
        lib = C.CDLL(os.path.join(BUILD, FLAVOURS[key][0]))
        lib.orc_forward.restype = C.c_int64
        lib.orc_backward.restype = C.c_int
        lib.orc_sizeof_real.restype = C.c_int
        assert lib.orc_sizeof_real() == size
        lib.orc_set_threads(_THREADS)
        _LIBS[key] = lib
    return _LIBS[key]


def _structs(real):
    class OrcIn(C.Structure):
        _fields_ = [("variant", C.c_int32), ("P", C.c_int32), ("D", C.c_int32), ("M", C.c_int32),
                    ("W", C.c_int32), ("H", C.c_int32), ("prefiltered", C.c_int32), ("pad_", C.c_int32),
                    ("tanfovx", real), ("tanfovy", real), ("scale_modifier", real),
                    ("gaze_x", real), ("gaze_y", real), ("alpha", real)] + \
                   [(n, C.c_void_p) for n in ("bg", "viewmatrix", "projmatrix", "campos", "means3D", "scales",
                                              "rotations", "opacities", "shs", "cov3D_precomp", "colors_precomp",
                                              "shs_dcs", "highest_levels", "loss_map")] + [("win", C.c_int32 * 4), ("cur_level", real)]

    class OrcOut(C.Structure):
        _fields_ = [(n, C.c_void_p) for n in ("depths", "radii", "means2D", "cov3D", "conic", "rgb", "clamped",
                                              "tiles_rect", "tiles_touched", "eigen_len", "eigen_vec",
                                              "level_ranges", "fov_colors", "tile_levels", "tile_gx", "tile_gy",
                                              "tile_min", "tile_blend", "ranges")] + \
                   [("capacity", C.c_int64), ("point_list", C.c_void_p), ("keys", C.c_void_p),
                    ("num_rect", C.c_int64), ("num_rendered", C.c_int64)] + \
                   [(n, C.c_void_p) for n in ("color", "final_T", "n_contrib", "gaussians_count", "contributions")]

    class OrcGrads(C.Structure):
        _fields_ = [(n, C.c_void_p) for n in ("dL_dpix", "dL_dmean2D", "dL_dconic", "dL_dopacity", "dL_dcolor",
                                              "dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot")]
    return OrcIn, OrcOut, OrcGrads


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _prep_inputs(variant, scene, cam, dtype, keep):
    """Build the orc_in struct. `keep` collects the converted arrays so they outlive the call."""
    real = C.c_float if np.dtype(dtype).itemsize == 4 else C.c_double
    OrcIn, OrcOut, OrcGrads = _structs(real)

    def arr(x):
        if x is None:
            return None
        a = np.ascontiguousarray(np.asarray(x, dtype=dtype))
        keep.append(a)
        return a

    means3D = arr(scene["means3D"])
    P = means3D.shape[0]
    shs = arr(scene.get("shs"))
    M = 0 if shs is None or shs.size == 0 else shs.shape[1]
    vi = VARIANTS[variant] if isinstance(variant, str) else int(variant)
    inp = OrcIn()
    inp.variant = vi
    inp.P, inp.D, inp.M = P, int(cam["sh_degree"]), M
    inp.W, inp.H = int(cam["image_width"]), int(cam["image_height"])
    inp.prefiltered = int(bool(cam.get("prefiltered", False)))
    inp.tanfovx, inp.tanfovy = float(cam["tanfovx"]), float(cam["tanfovy"])
    inp.scale_modifier = float(cam.get("scale_modifier", 1.0))
    gaze = cam.get("gaze", (0.5, 0.5))
    inp.gaze_x, inp.gaze_y = float(np.float32(gaze[0])), float(np.float32(gaze[1]))
    inp.alpha = float(np.float32(cam.get("alpha", 0.05)))
    inp.bg = _ptr(arr(cam["bg"]))
    inp.viewmatrix = _ptr(arr(cam["viewmatrix"]).reshape(-1))
    inp.projmatrix = _ptr(arr(cam["projmatrix"]).reshape(-1))
    inp.campos = _ptr(arr(cam["campos"]))
    inp.means3D = _ptr(means3D)
    inp.scales = _ptr(arr(scene.get("scales")))
    inp.rotations = _ptr(arr(scene.get("rotations")))
    inp.opacities = _ptr(arr(scene["opacities"]))
    inp.shs = _ptr(shs if M else None)
    inp.cov3D_precomp = _ptr(arr(scene.get("cov3D_precomp")))
    inp.colors_precomp = _ptr(arr(scene.get("colors_precomp")))
    inp.shs_dcs = _ptr(arr(scene.get("shs_dcs")))
    inp.highest_levels = _ptr(arr(scene.get("highest_levels")))
    inp.loss_map = _ptr(arr(scene.get("loss_map")))
    inp.cur_level = float(cam.get("cur_level", 0.0))
    win = cam.get("tile_window")  # (x0, y0, x1, y1) in tiles; bench cpu_baseline sampling only
    if win is not None:
        for i in range(4):
            inp.win[i] = int(win[i])
    return inp, OrcOut, OrcGrads, P, M, vi


def _alloc_outputs(OrcOut, P, W, H, dtype, capacity):
    T = ((W + 15) // 16) * ((H + 15) // 16)
    f = lambda *s: np.zeros(s, dtype=dtype)
    o = {
        "depths": f(P), "radii": np.zeros(P, np.int32), "means2D": f(P, 2), "cov3D": f(P, 6), "conic": f(P, 3),
        "rgb": f(P, 3), "clamped": np.zeros((P, 3), np.uint8), "tiles_rect": np.zeros(P, np.uint32),
        "tiles_touched": np.zeros(P, np.uint32), "eigen_len": f(P, 2), "eigen_vec": f(P, 4),
        "level_ranges": np.zeros((P, 2), np.int32), "fov_colors": np.full((P, FOV_NUM, 3), np.nan, dtype=dtype),
        "tile_levels": f(T), "tile_gx": f(T), "tile_gy": f(T), "tile_min": f(T),
        "tile_blend": np.zeros(T, np.uint8), "ranges": np.zeros((T, 2), np.uint32),
        "point_list": np.zeros(max(capacity, 1), np.uint32), "keys": np.zeros(max(capacity, 1), np.uint64),
        "color": f(3, H, W), "final_T": f(H, W), "n_contrib": np.zeros((H, W), np.uint32),
        "gaussians_count": np.zeros(P, np.int32), "contributions": f(P),
    }
    out = OrcOut()
    for k, v in o.items():
        setattr(out, k, _ptr(v))
    out.capacity = capacity
    return out, o


def forward(variant, scene, cam, dtype=np.float32, fma=False):
    """Run the oracle forward. Returns a dict with every stage's outputs. fma: the contracted float flavour (see FLAVOURS)."""
    lib = _lib(dtype, fma)
    keep = []
    inp, OrcOut, _, P, M, vi = _prep_inputs(variant, scene, cam, dtype, keep)
    W, H = inp.W, inp.H
    capacity = int(cam.get("capacity_hint", 0))  # saves the second pass when the caller knows a bound on the instances
    for _ in range(2):
        out, o = _alloc_outputs(OrcOut, P, W, H, dtype, capacity)
        n = lib.orc_forward(C.byref(inp), C.byref(out))
        if n < 0:
            raise RuntimeError("oracle: prefiltered point was culled (reference would __trap)")
        if n <= capacity:
            break
        capacity = int(n)
    o["num_rendered"] = int(out.num_rendered)
    o["num_rect"] = int(out.num_rect)
    o["point_list"] = o["point_list"][:o["num_rendered"]]
    o["keys"] = o["keys"][:o["num_rendered"]]
    o["variant"] = vi
    return o


def backward(variant, scene, cam, fwd, dL_dpix, dtype=np.float32, fma=False):
    """Run the oracle backward (R0 / RS only) given the dict returned by forward()."""
    lib = _lib(dtype, fma)
    keep = []
    inp, OrcOut, OrcGrads, P, M, vi = _prep_inputs(variant, scene, cam, dtype, keep)
    out = OrcOut()
    for k in ("depths", "radii", "means2D", "cov3D", "conic", "rgb", "clamped", "tiles_rect", "tiles_touched",
              "eigen_len", "eigen_vec", "ranges", "point_list", "final_T", "n_contrib"):
        a = np.ascontiguousarray(fwd[k])
        keep.append(a)
        setattr(out, k, _ptr(a))
    out.capacity = fwd["num_rendered"]
    out.num_rendered = fwd["num_rendered"]
    f = lambda *s: np.zeros(s, dtype=dtype)
    g = {"dL_dmean2D": f(P, 3), "dL_dconic": f(P, 2, 2), "dL_dopacity": f(P, 1), "dL_dcolor": f(P, 3),
         "dL_dmean3D": f(P, 3), "dL_dcov3D": f(P, 6), "dL_dsh": f(P, max(M, 0), 3), "dL_dscale": f(P, 3),
         "dL_drot": f(P, 4)}
    gs = OrcGrads()
    dpix = np.ascontiguousarray(np.asarray(dL_dpix, dtype=dtype))
    gs.dL_dpix = _ptr(dpix)
    for k, v in g.items():
        setattr(gs, k, _ptr(v))
    rc = lib.orc_backward(C.byref(inp), C.byref(out), C.byref(gs))
    if rc != 0:
        raise RuntimeError(f"oracle backward failed rc={rc}")
    return g


def tile_levels(cam, dtype=np.float32):
    """RF tile level map only (levels, gx, gy, tile_min, blend flags)."""
    lib = _lib(dtype)
    keep = []
    scene = {"means3D": np.zeros((0, 3)), "opacities": np.zeros((0, 4))}
    inp, OrcOut, _, P, M, vi = _prep_inputs("fov_pcheck_obb", scene, cam, dtype, keep)
    out, o = _alloc_outputs(OrcOut, 0, inp.W, inp.H, dtype, 0)
    lib.orc_tile_levels(C.byref(inp), C.byref(out))
    return {k: o[k] for k in ("tile_levels", "tile_gx", "tile_gy", "tile_min", "tile_blend")}


def sh_colors(scene, cam, rest=False, dtype=np.float32):
    """Unclamped SH colour (+0.5) per Gaussian; rest=True evaluates only degrees 1..3."""
    lib = _lib(dtype)
    keep = []
    inp, _, _, P, M, vi = _prep_inputs("original", scene, cam, dtype, keep)
    out = np.zeros((P, 3), dtype=dtype)
    lib.orc_sh_colors(C.byref(inp), C.c_int(int(rest)), _ptr(out))
    return out


def mark_visible(scene, cam, dtype=np.float32):
    lib = _lib(dtype)
    keep = []
    inp, _, _, P, M, vi = _prep_inputs("original", scene, cam, dtype, keep)
    out = np.zeros(P, np.uint8)
    lib.orc_mark_visible(C.byref(inp), _ptr(out))
    return out.astype(bool)
