"""A second, INDEPENDENT derivation of the OBB-culled and the foveated forward rasterizers, in numpy.

TEST INFRASTRUCTURE ONLY (lives under tests/; never imported by the product package, never on a GPU path). Written from the
reference's CUDA sources alone -- not from oracle/fovraster_oracle.c -- so that the one hand restatement the GPU parity tests
lean on has a twin that was derived separately (tests/test_second_derivation.py compares the two: `radii`, `ranges`,
`point_list`, level ranges array-equal in float32, images in float64). Files followed (paths under
/root/reference/fov3dgs/submodules/, RS = diff-gaussian-rasterization_pcheck_obb_sum, RP = ..._pcheck_obb,
RF = ..._fov_pcheck_obb):

  per-Gaussian projection, conic, radius, OBB axes        RF cuda_rasterizer/forward.cu:22-238 (RS forward.cu:74-293)
  glm products (column-major, three-term sums)             third_party/glm/glm/detail/type_mat3x3.inl:486-520
  transformPoint*, ndc2Pix, getRect, in_frustum            RF cuda_rasterizer/auxiliary.h:168-215,271-296
  OBB_check, ps2level, normalize (rsqrtf)                  RF cuda_rasterizer/auxiliary.h:55-166
  OBB_test (RS / RP)                                       RS cuda_rasterizer/rasterizer_impl.cu:70-146
  tile levels / level infos                                RF cuda_rasterizer/rasterizer_impl.cu:86-260
  filter, level_ranges                                     RF cuda_rasterizer/rasterizer_impl.cu:264-383
  keys, stable radix sort, tile ranges                     RF cuda_rasterizer/rasterizer_impl.cu:423-486,849-880
  per-level colours                                        RF cuda_rasterizer/rasterizer_impl.cu:37-84,490-530
  full SH colour (RS / RP)                                 RS cuda_rasterizer/forward.cu:20-72
  blend: single level / two levels                         RF cuda_rasterizer/forward.cu:490-609 / :262-476
  blend RS (counts per fetched round, contributions)       RS cuda_rasterizer/forward.cu:298-430

Arithmetic: every expression is evaluated in the order the C++ source gives it, in `dtype` (float32 = the reference's
arithmetic, including the places where a double literal promotes an expression: ndc2Pix, the pooling angles, M_PI, 3.9;
float64 = everything in double). No multiply-add is fused (the reference's binary was built with nvcc's default
-fmad=true: see tests/test_second_derivation.py for what that can move). CUDA's `rsqrtf` (2 ulp) is taken as 1/sqrt;
acosf / tanf come from the host's libm through ctypes so that float32 tile levels have ONE definition on this host.
"""
import ctypes
import ctypes.util

import numpy as np

FOV_NUM = 4
REAL_IMAGE_WIDTH = 2.0
REAL_VIEWING_DISTANCE = 1.0
SQRT_MAX_PS = 3.4641016151377544
START_BLEND = 0.5
BLEND_WIDTH = 0.5
BLOCK = 16
BLOCK_SIZE = 256

_libm = ctypes.CDLL(ctypes.util.find_library("m"))
for _n in ("acosf", "tanf"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float]
for _n in ("acos", "tan"):
    getattr(_libm, _n).restype = ctypes.c_double
    getattr(_libm, _n).argtypes = [ctypes.c_double]


class _Arith:
    """Scalars and helpers of one precision."""

    def __init__(self, dtype):
        self.F = np.dtype(dtype).type
        self.single = np.dtype(dtype).itemsize == 4
        # a float literal of the source (1.3f, 0.3f, SH_C1 ...) IS a float: round it once, then widen for the double build
        self.lit = (lambda v: self.F(np.float32(v)))

    def acos(self, x):
        return self.F(_libm.acosf(float(x))) if self.single else self.F(_libm.acos(float(x)))

    def tan(self, x):
        return self.F(_libm.tanf(float(x))) if self.single else self.F(_libm.tan(float(x)))

    def via_double(self, x):
        """An expression the source evaluates in double, stored to a float."""
        return self.F(x)


def _sum3(a, b, c):
    return (a + b) + c


def _glm_mul(A, B):
    """glm::mat3 * glm::mat3 on arrays [..., col, row]: Result[c][r] = A[0][r] B[c][0] + A[1][r] B[c][1] + A[2][r] B[c][2]."""
    R = np.empty_like(A)
    for c in range(3):
        for r in range(3):
            R[..., c, r] = _sum3(A[..., 0, r] * B[..., c, 0], A[..., 1, r] * B[..., c, 1], A[..., 2, r] * B[..., c, 2])
    return R


def _glm_t(A):
    return np.swapaxes(A, -1, -2).copy()


def _xform4x4(p, m):
    return [((m[0 + i] * p[:, 0] + m[4 + i] * p[:, 1]) + m[8 + i] * p[:, 2]) + m[12 + i] for i in range(4)]


def _sh_terms(ar, deg, sh, d, first):
    """The degree >= 1 part of computeColorFromSH / computeRestColorFromSH. sh[:, first + k - 1] is the source's sh[k]."""
    lit = ar.lit
    C1 = lit(0.4886025119029199)
    C2 = [lit(v) for v in (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)]
    C3 = [lit(v) for v in (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
                           1.445305721320277, -0.5900435899266435)]
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    s = lambda k: sh[:, first + k - 1]
    two, three, four = ar.F(2), ar.F(3), ar.F(4)

    def add(res):
        if deg > 0:
            res = res - (C1 * y) * s(1) + (C1 * z) * s(2) - (C1 * x) * s(3)
            if deg > 1:
                xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
                res = (res + (C2[0] * xy) * s(4) + (C2[1] * yz) * s(5) + (C2[2] * (two * zz - xx - yy)) * s(6)
                       + (C2[3] * xz) * s(7) + (C2[4] * (xx - yy)) * s(8))
                if deg > 2:
                    res = (res + ((C3[0] * y) * (three * xx - yy)) * s(9) + ((C3[1] * xy) * z) * s(10)
                           + ((C3[2] * y) * (four * zz - xx - yy)) * s(11)
                           + ((C3[3] * z) * (two * zz - three * xx - three * yy)) * s(12)
                           + ((C3[4] * x) * (four * zz - xx - yy)) * s(13) + ((C3[5] * z) * (xx - yy)) * s(14)
                           + ((C3[6] * x) * (xx - three * yy)) * s(15))
        return res
    return add


def _view_dirs(means, campos):
    d = means - campos[None, :]
    length = np.sqrt(_sum3(d[:, 0] * d[:, 0], d[:, 1] * d[:, 1], d[:, 2] * d[:, 2]))
    return d / length[:, None]


def project(ar, scene, cam):
    """preprocessCUDA up to the OBB axes. Returns per-Gaussian arrays; radii == 0 marks the dropped ones."""
    F, lit = ar.F, ar.lit
    W, H = int(cam["image_width"]), int(cam["image_height"])
    means = np.asarray(scene["means3D"], F)
    scales = np.asarray(scene["scales"], F)
    rot = np.asarray(scene["rotations"], F)
    vm = np.asarray(cam["viewmatrix"], F).reshape(-1)
    pm = np.asarray(cam["projmatrix"], F).reshape(-1)
    tanx, tany = F(np.float32(cam["tanfovx"])), F(np.float32(cam["tanfovy"]))
    mod = F(np.float32(cam.get("scale_modifier", 1.0)))
    P = means.shape[0]
    gx, gy = (W + BLOCK - 1) // BLOCK, (H + BLOCK - 1) // BLOCK
    focal_y = F(H) / (F(2) * tany)
    focal_x = F(W) / (F(2) * tanx)
    with np.errstate(all="ignore"):
        ph = _xform4x4(means, pm)
        p_w = F(1) / (ph[3] + lit(0.0000001))
        proj = [ph[0] * p_w, ph[1] * p_w, ph[2] * p_w]
        t = _xform4x4(means, vm)[:3]
        in_front = ~(t[2] <= lit(0.2))
        # computeCov3D
        r, x, y, z = rot[:, 0], rot[:, 1], rot[:, 2], rot[:, 3]
        one, two = F(1), F(2)
        a = [one - two * (y * y + z * z), two * (x * y - r * z), two * (x * z + r * y),
             two * (x * y + r * z), one - two * (x * x + z * z), two * (y * z - r * x),
             two * (x * z - r * y), two * (y * z + r * x), one - two * (x * x + y * y)]
        R = np.stack(a, 1).reshape(P, 3, 3)                       # R[c][r] = a[3c + r]
        S = np.zeros((P, 3, 3), F)
        for i in range(3):
            S[:, i, i] = mod * scales[:, i]
        M = _glm_mul(S, R)
        Sigma = _glm_mul(_glm_t(M), M)
        cov3D = np.stack([Sigma[:, 0, 0], Sigma[:, 0, 1], Sigma[:, 0, 2], Sigma[:, 1, 1], Sigma[:, 1, 2], Sigma[:, 2, 2]], 1)
        # computeCov2D
        limx, limy = lit(1.3) * tanx, lit(1.3) * tany
        txtz, tytz = t[0] / t[2], t[1] / t[2]
        tx = np.fmin(limx, np.fmax(-limx, txtz)) * t[2]
        ty = np.fmin(limy, np.fmax(-limy, tytz)) * t[2]
        tz = t[2]
        J = np.zeros((P, 3, 3), F)
        J[:, 0, 0] = focal_x / tz
        J[:, 0, 2] = -(focal_x * tx) / (tz * tz)
        J[:, 1, 1] = focal_y / tz
        J[:, 1, 2] = -(focal_y * ty) / (tz * tz)
        Wm = np.empty((P, 3, 3), F)
        for c in range(3):
            for rr in range(3):
                Wm[:, c, rr] = vm[4 * rr + c]                     # W[0] = (vm0, vm4, vm8), W[1] = (vm1, vm5, vm9) ...
        Tm = _glm_mul(Wm, J)
        Vrk = np.empty((P, 3, 3), F)
        sym = ((0, 1, 2), (1, 3, 4), (2, 4, 5))
        for c in range(3):
            for rr in range(3):
                Vrk[:, c, rr] = cov3D[:, sym[c][rr]]
        cov = _glm_mul(_glm_mul(_glm_t(Tm), _glm_t(Vrk)), Tm)
        ca = cov[:, 0, 0] + lit(0.3)
        cb = cov[:, 0, 1]
        cc = cov[:, 1, 1] + lit(0.3)
        det = ca * cc - cb * cb
        ok = in_front & ~(det == 0)
        det_inv = F(1) / det
        conic = np.stack([cc * det_inv, -cb * det_inv, ca * det_inv], 1)
        mid = F(0.5) * (ca + cc)
        root = np.sqrt(np.fmax(lit(0.1), mid * mid - det))
        lam1, lam2 = mid + root, mid - root
        my_radius = np.ceil(F(3) * np.sqrt(np.fmax(lam1, lam2)))
        # ndc2Pix: ((v + 1.0) * S - 1.0) * 0.5 -- double literals
        px = ((proj[0].astype(np.float64) + 1.0) * W - 1.0) * 0.5
        py = ((proj[1].astype(np.float64) + 1.0) * H - 1.0) * 0.5
        pix = np.stack([px.astype(F), py.astype(F)], 1)
        radius_i = np.where(ok & np.isfinite(my_radius), my_radius, 0).astype(np.int64)   # float -> int argument of getRect
        rect = get_rect(ar, pix, radius_i, gx, gy)
        tnum = (rect[:, 3] - rect[:, 1]) * (rect[:, 2] - rect[:, 0])
        ok = ok & (tnum > 0)
        # OBB axes (only for rects of more than one tile)
        a1, a2, b1 = ca - lam1, ca - lam2, cb
        v1 = np.stack([-b1, a1], 1)
        v2 = np.stack([-b1, a2], 1)
        n1 = F(1) / np.sqrt(v1[:, 0] * v1[:, 0] + v1[:, 1] * v1[:, 1])
        n2 = F(1) / np.sqrt(v2[:, 0] * v2[:, 0] + v2[:, 1] * v2[:, 1])
        v1 = v1 * n1[:, None]
        v2 = v2 * n2[:, None]
        len1 = F(3) * np.sqrt(lam1)
        len2 = F(3) * np.sqrt(lam2)
        multi = ok & (tnum > 1)
        v1[~multi] = 0
        v2[~multi] = 0
        len1 = np.where(multi, len1, F(0))
        len2 = np.where(multi, len2, F(0))
    radii = np.where(ok, radius_i, 0).astype(np.int32)
    return dict(radii=radii, depth=t[2], pix=pix, conic=conic, tnum=np.where(ok, tnum, 0), v1=v1, v2=v2, len1=len1, len2=len2,
                cov3D=cov3D, grid=(gx, gy))


def get_rect(ar, pix, radius_i, gx, gy):
    """getRect: C (int) truncation of float quotients, clamped to the grid. -> [P, 4] = x0, y0, x1, y1."""
    F = ar.F
    rf = radius_i.astype(F)                                       # float - int: the int is converted to float
    with np.errstate(all="ignore"):
        q = [(pix[:, 0] - rf) / F(BLOCK), (pix[:, 1] - rf) / F(BLOCK),
             (pix[:, 0] + rf + F(BLOCK - 1)) / F(BLOCK), (pix[:, 1] + rf + F(BLOCK - 1)) / F(BLOCK)]
    out = np.empty((pix.shape[0], 4), np.int64)
    for i, (v, lim) in enumerate(zip(q, (gx, gy, gx, gy))):
        v = np.where(np.isfinite(v), v, 0)
        v = np.clip(v, -2.0e9, 2.0e9)
        out[:, i] = np.minimum(lim, np.maximum(0, np.trunc(v).astype(np.int64)))
    return out


def obb_pairs(ar, pr, extra_keep=None):
    """Walk every visible Gaussian's tile rectangle (y outer, x inner) and apply OBB_check to the rectangles of more than one
    tile. `extra_keep(g_of_pair, tile_of_pair) -> bool[pairs]` is RF's level test, evaluated BEFORE the box test as in `filter`.
    Returns (g, tile, kept) per pair in emission order."""
    F = ar.F
    gx, gy = pr["grid"]
    vis = np.nonzero(pr["radii"] > 0)[0]
    rect = get_rect(ar, pr["pix"][vis], pr["radii"][vis].astype(np.int64), gx, gy)
    wx, wy = rect[:, 2] - rect[:, 0], rect[:, 3] - rect[:, 1]
    n = wx * wy
    g = np.repeat(vis, n)
    first = np.repeat(np.cumsum(n) - n, n)
    k = np.arange(n.sum()) - first
    wxr = np.repeat(wx, n)
    tx = np.repeat(rect[:, 0], n) + k % wxr
    ty = np.repeat(rect[:, 1], n) + k // wxr
    tile = ty * gx + tx
    single = np.repeat(n == 1, n)
    keep = np.ones(g.shape[0], bool) if extra_keep is None else extra_keep(g, tile)
    cx, cy = pr["pix"][g, 0], pr["pix"][g, 1]
    v1, v2, l1, l2 = pr["v1"][g], pr["v2"][g], pr["len1"][g], pr["len2"][g]
    with np.errstate(all="ignore"):
        d1x, d1y, d2x, d2y = l1 * v1[:, 0], l1 * v1[:, 1], l2 * v2[:, 0], l2 * v2[:, 1]
        vx = [cx + d1x + d2x, cx - d1x + d2x, cx - d1x - d2x, cx + d1x - d2x]
        vy = [cy + d1y + d2y, cy - d1y + d2y, cy - d1y - d2y, cy + d1y - d2y]
        tpx = tx.astype(F) * F(BLOCK) + F(BLOCK) / F(2)
        tpy = ty.astype(F) * F(BLOCK) + F(BLOCK) / F(2)
        eight = F(8)

        def span(vals):
            lo = hi = vals[0]
            for v in vals[1:]:
                lo, hi = np.fmin(lo, v), np.fmax(hi, v)
            return lo, hi
        lo, hi = span([v - tpx for v in vx])
        out = (hi < -eight) | (lo > eight)
        lo, hi = span([v - tpy for v in vy])
        out |= (hi < -eight) | (lo > eight)
        corners = [(tpx + eight - cx, tpy + eight - cy), (tpx - eight - cx, tpy + eight - cy),
                   (tpx - eight - cx, tpy - eight - cy), (tpx + eight - cx, tpy - eight - cy)]
        for v, ln in ((v1, l1), (v2, l2)):
            lo, hi = span([c[0] * v[:, 0] + c[1] * v[:, 1] for c in corners])
            out |= (ln < lo) | (-ln > hi)
    kept = keep & (single | ~out)
    return g, tile, kept


def tile_level_map(ar, cam):
    """compute_tile_levels_cuda + compute_tile_level_infos_cuda."""
    F, lit = ar.F, ar.lit
    W, H = int(cam["image_width"]), int(cam["image_height"])
    gxn, gyn = (W + 15) // BLOCK, (H + 15) // 16
    T = gxn * gyn
    gaze = cam.get("gaze", (0.5, 0.5))
    gaze = (F(np.float32(gaze[0])), F(np.float32(gaze[1])))
    alpha = F(np.float32(cam.get("alpha", 0.05)))
    riw, rvd = F(REAL_IMAGE_WIDTH), F(REAL_VIEWING_DISTANCE)
    rih = F(H) / F(W) * riw

    def dist(x, y, z):
        return np.sqrt(F(_sum3(x * x, y * y, z * z)))

    def ncd2dir(nx, ny):
        v = [(nx - F(0.5)) * riw, (ny - F(0.5)) * rih, rvd]
        d = dist(*v)
        return [v[0] / d, v[1] / d, v[2] / d]

    def dot(a, b):
        return _sum3(a[0] * b[0], a[1] * b[1], a[2] * b[2])
    step = F((float(lit(SQRT_MAX_PS)) - 1.0) / float(F(FOV_NUM - 1)))          # double expression -> const float
    cap = float(F(FOV_NUM)) - 0.1                                              # double
    gaze_dir, centre_dir = ncd2dir(*gaze), ncd2dir(F(0.5), F(0.5))
    levels = np.empty(T, F)
    with np.errstate(all="ignore"):
        for idx in range(T):
            ty_, tx_ = divmod(idx, gxn)
            px, py = F(tx_ * BLOCK + BLOCK // 2), F(ty_ * BLOCK + BLOCK // 2)
            nx, ny = px / F(W), py / F(H)
            tdir = ncd2dir(nx, ny)
            ecc = ar.acos(dot(gaze_dir, tdir))
            ecc_c = ar.acos(dot(tdir, centre_dir))
            pooling = alpha * ecc * ecc
            a_min = ar.via_double(float(ecc_c) - float(pooling) * 0.5)
            a_max = ar.via_double(float(ecc_c) + float(pooling) * 0.5)
            d2p = dist(F((float(nx) - 0.5) * float(riw)), F((float(ny) - 0.5) * float(rih)), rvd)
            major = (ar.tan(a_max) - ar.tan(a_min)) * rvd
            minor = F(2) * d2p * ar.tan(pooling * F(0.5))
            area = ar.via_double(np.pi * float(major) * float(minor) * 0.25)
            ps = np.sqrt(area) * (F(W) / riw)
            if ps <= 1:
                lv = F(0)
            else:
                lv = (np.sqrt(ps) - F(1)) / step
            if float(lv) > cap:
                lv = F(cap)
            levels[idx] = lv
    gxs, gys, tmin = np.zeros(T, F), np.zeros(T, F), np.empty(T, F)
    blend = np.zeros(T, bool)
    minus1 = F(-1)
    for idx in range(T):
        ty_, tx_ = divmod(idx, gxn)
        lv = levels[idx]
        right = levels[idx + 1] if tx_ + 1 < gxn else minus1
        left = levels[idx - 1] if tx_ - 1 >= 0 else minus1
        up = levels[idx + gxn] if ty_ + 1 < gyn else minus1
        down = levels[idx - gxn] if ty_ - 1 >= 0 else minus1
        g_x = g_y = F(0)
        if right != -1 and left != -1:
            g_x = (right - left) / F(2)
        elif right != -1:
            g_x = right - lv
        elif left != -1:
            g_x = lv - left
        if up != -1 and down != -1:
            g_y = (up - down) / F(2)
        elif up != -1:
            g_y = up - lv
        elif down != -1:
            g_y = lv - down
        max_delta = ar.via_double(0.5 * float(abs(g_x) + abs(g_y)))
        tm = lv - max_delta
        tm_i = F(int(tm)) if np.isfinite(tm) else F(0)
        blend[idx] = (tm - tm_i) > F(START_BLEND) and tm_i < (FOV_NUM - 1)
        tmin[idx], gxs[idx], gys[idx] = tm, g_x, g_y
    return dict(tile_levels=levels, tile_gx=gxs, tile_gy=gys, tile_min=tmin, tile_blend=blend)


def sort_instances(pr, g, tile, kept, T):
    """duplicateWithKeys + the stable radix sort on (tile, depth bits) + identifyTileRanges."""
    g, tile = g[kept], tile[kept]
    depth32 = pr["depth"].astype(np.float32)                       # the key holds the bits of a float
    bits = depth32.view(np.uint32).astype(np.uint64)
    keys = (tile.astype(np.uint64) << np.uint64(32)) | bits[g]
    order = np.argsort(keys, kind="stable")                        # emission order = Gaussian index order, kept for equal keys
    point_list = g[order].astype(np.uint32)
    st = tile[order]
    ranges = np.zeros((T, 2), np.uint32)
    if st.size:
        starts = np.searchsorted(st, np.arange(T), side="left")
        ends = np.searchsorted(st, np.arange(T), side="right")
        has = ends > starts
        ranges[has, 0] = starts[has]
        ranges[has, 1] = ends[has]
    return point_list, ranges, keys[order]


def _tile_pixels(W, H, gx, gy):
    """Per tile: pixel coordinates of its 256 threads (thread_rank = 16 y + x) and the `inside` mask."""
    t = np.arange(gx * gy)
    ox, oy = (t % gx) * BLOCK, (t // gx) * BLOCK
    lx, ly = np.tile(np.arange(BLOCK), BLOCK), np.repeat(np.arange(BLOCK), BLOCK)
    X, Y = ox[:, None] + lx[None, :], oy[:, None] + ly[None, :]
    return X, Y, (X < W) & (Y < H), lx, ly


def _power(ar, conic, mx, my, pxf, pyf):
    dx, dy = mx - pxf, my - pyf
    return -ar.F(0.5) * (conic[:, 0:1] * dx * dx + conic[:, 2:3] * dy * dy) - conic[:, 1:2] * dx * dy


def _exp(ar, p):
    if ar.single:   # expf: correctly rounded from the double result (glibc's expf is, to 0.502 ulp)
        return np.exp(p.astype(np.float64)).astype(np.float32)
    return np.exp(p)


def blend_plain(ar, pr, cam, point_list, ranges, opac, rgb, stats):
    """RS / RP renderCUDA: all tiles at once, one list position per step. stats: gaussians_count / contributions (RS)."""
    F, lit = ar.F, ar.lit
    W, H = int(cam["image_width"]), int(cam["image_height"])
    gx, gy = pr["grid"]
    X, Y, inside, _, _ = _tile_pixels(W, H, gx, gy)
    pxf, pyf = X.astype(F), Y.astype(F)
    nt = gx * gy
    Tr = np.ones((nt, BLOCK_SIZE), F)
    C = np.zeros((nt, BLOCK_SIZE, 3), F)
    done = ~inside
    contributor = np.zeros((nt, BLOCK_SIZE), np.int64)
    last = np.zeros((nt, BLOCK_SIZE), np.int64)
    start = ranges[:, 0].astype(np.int64)
    lens = ranges[:, 1].astype(np.int64) - start
    alive = lens > 0
    P = pr["radii"].shape[0]
    gcount = np.zeros(P, np.int64)
    contrib = np.zeros(P, np.float64)
    conic = pr["conic"]
    for j in range(int(lens.max()) if nt else 0):
        if j % BLOCK_SIZE == 0:                                    # __syncthreads_count(done) == BLOCK_SIZE -> break
            alive = alive & ~done.all(1) & (lens > j)
            if stats:
                for t in np.nonzero(alive)[0]:
                    np.add.at(gcount, point_list[start[t] + j: start[t] + min(j + BLOCK_SIZE, lens[t])], 1)
        ti = np.nonzero(alive & (lens > j))[0]
        if ti.size == 0:
            break
        g = point_list[start[ti] + j].astype(np.int64)
        live = ~done[ti]
        contributor[ti] += live
        power = _power(ar, conic[g], pr["pix"][g, 0:1], pr["pix"][g, 1:2], pxf[ti], pyf[ti])
        with np.errstate(all="ignore"):
            go = live & ~((power > 0) | (power < lit(-4.5)))
            alpha = np.fmin(lit(0.99), opac[g][:, None] * _exp(ar, power))
            go &= ~(alpha < F(1) / F(255))
            test_T = Tr[ti] * (F(1) - alpha)
            stop = go & (test_T < lit(0.0001))
            acc = go & ~stop
            w = alpha * Tr[ti]
            if stats:                                              # C += features * alpha * T, left to right (RS forward.cu:404)
                add = (rgb[g][:, None, :] * alpha[..., None]) * Tr[ti][..., None]
                np.add.at(contrib, g, np.where(acc, w, 0).astype(np.float64).sum(1))
            else:                                                  # RP: w = alpha * T; C += feature * w
                add = rgb[g][:, None, :] * w[..., None]
        C[ti] = np.where(acc[..., None], C[ti] + add, C[ti])
        Tr[ti] = np.where(acc, test_T, Tr[ti])
        last[ti] = np.where(acc, contributor[ti], last[ti])
        done[ti] |= stop
    bg = np.asarray(cam["bg"], F)
    out = C + Tr[..., None] * bg[None, None, :]
    img = np.zeros((3, H, W), F)
    fT = np.zeros((H, W), F)
    nc = np.zeros((H, W), np.uint32)
    for c in range(3):
        img[c][Y[inside], X[inside]] = out[..., c][inside]
    fT[Y[inside], X[inside]] = Tr[inside]
    nc[Y[inside], X[inside]] = last[inside]
    return dict(color=img, final_T=fT, n_contrib=nc, gaussians_count=gcount.astype(np.int32), contributions=contrib.astype(F))


def blend_fov(ar, pr, cam, lv, point_list, ranges, opac4, colours, highest):
    """RF renderCUDA (single-level tiles) and renderCUDA_blending (two-level tiles); both consume tile_level_MIN."""
    F, lit = ar.F, ar.lit
    W, H = int(cam["image_width"]), int(cam["image_height"])
    gx, gy = pr["grid"]
    X, Y, inside, lx, ly = _tile_pixels(W, H, gx, gy)
    pxf, pyf = X.astype(F), Y.astype(F)
    nt = gx * gy
    tmin, blend = lv["tile_min"], lv["tile_blend"]
    with np.errstate(all="ignore"):
        L1 = np.where(np.isfinite(tmin), tmin, 0).astype(np.int64)   # (int)tile_level_f
    L2 = L1 + 1
    est = tmin[:, None] + (lx.astype(F)[None, :] * lv["tile_gx"][:, None] + ly.astype(F)[None, :] * lv["tile_gy"][:, None]) / F(BLOCK)
    L2f = tmin + F(1)
    T1 = np.ones((nt, BLOCK_SIZE), F)
    T2 = np.ones((nt, BLOCK_SIZE), F)
    C1 = np.zeros((nt, BLOCK_SIZE, 3), F)
    C2 = np.zeros((nt, BLOCK_SIZE, 3), F)
    done = ~inside
    L1_done = blend[:, None] & (est > L2.astype(F)[:, None])
    L2_done = np.broadcast_to(~blend[:, None], (nt, BLOCK_SIZE)).copy()   # single-level tiles never touch the second state
    start = ranges[:, 0].astype(np.int64)
    lens = ranges[:, 1].astype(np.int64) - start
    conic = pr["conic"]
    for j in range(int(lens.max()) if nt else 0):
        ti = np.nonzero(lens > j)[0]
        if ti.size == 0:
            break
        g = point_list[start[ti] + j].astype(np.int64)
        isb = blend[ti][:, None]
        power = _power(ar, conic[g], pr["pix"][g, 0:1], pr["pix"][g, 1:2], pxf[ti], pyf[ti])
        with np.errstate(all="ignore"):
            go = ~done[ti] & ~((power > 0) | (power < lit(-4.5)))
            e = _exp(ar, power)
            # level 1 (the only level of a single-level tile)
            a1 = np.fmin(lit(0.99), opac4[g, np.minimum(L1[ti], FOV_NUM - 1)][:, None] * e)
            go1 = go & ~L1_done[ti] & ~(a1 < F(1) / F(255))
            t1 = T1[ti] * (F(1) - a1)
            stop1 = go1 & (t1 < lit(0.0001))
            acc1 = go1 & ~stop1
            w1 = a1 * T1[ti]
            f1 = colours[g, np.minimum(L1[ti], FOV_NUM - 1)][:, None, :]
            C1[ti] = np.where(acc1[..., None], C1[ti] + f1 * w1[..., None], C1[ti])
            T1[ti] = np.where(acc1, t1, T1[ti])
            L1_done[ti] |= stop1
            # level 2 (two-level tiles only)
            a2 = np.fmin(lit(0.99), opac4[g, np.minimum(L2[ti], FOV_NUM - 1)][:, None] * e)
            skip2 = (a2 < F(1) / F(255)) | ((highest[g] + F(1))[:, None] < L2f[ti][:, None])
            go2 = go & isb & ~L2_done[ti] & ~skip2
            t2 = T2[ti] * (F(1) - a2)
            stop2 = go2 & (t2 < lit(0.0001))
            acc2 = go2 & ~stop2
            w2 = a2 * T2[ti]
            f2 = colours[g, np.minimum(L2[ti], FOV_NUM - 1)][:, None, :]
            C2[ti] = np.where(acc2[..., None], C2[ti] + f2 * w2[..., None], C2[ti])
            T2[ti] = np.where(acc2, t2, T2[ti])
            L2_done[ti] |= stop2
        # single-level tile: the pixel is done when its one level is; two-level: when both are (checked only by pixels
        # that got past the support test of this entry -- nothing changed for the others)
        done[ti] |= go & L1_done[ti] & L2_done[ti]
    bg = np.asarray(cam["bg"], F)
    c1 = C1 + bg[None, None, :] * T1[..., None]
    c2 = C2 + bg[None, None, :] * T2[..., None]
    with np.errstate(all="ignore"):
        x = np.abs(est - (L1.astype(F)[:, None] + F(START_BLEND))) / F(BLEND_WIDTH)
        x = np.fmax(F(0), np.fmin(F(1), x))
        blend_T = F(3) * x * x - F(2) * x * x * x
        w = F(1) - blend_T
    two = c1 * w[..., None] + c2 * (F(1) - w)[..., None]
    out = np.where(blend[:, None, None], two, c1)
    img = np.zeros((3, H, W), F)
    for c in range(3):
        img[c][Y[inside], X[inside]] = out[..., c][inside]
    return dict(color=img)


def rasterize(variant, scene, cam, dtype=np.float32):
    """variant: "pcheck_obb", "pcheck_obb_sum" or "fov_pcheck_obb". scene / cam: the dicts of tests/helpers.py."""
    ar = _Arith(dtype)
    F, lit = ar.F, ar.lit
    pr = project(ar, scene, cam)
    gx, gy = pr["grid"]
    T = gx * gy
    P = pr["radii"].shape[0]
    means = np.asarray(scene["means3D"], F)
    campos = np.asarray(cam["campos"], F)
    deg = int(cam["sh_degree"])
    out = {}
    if variant == "fov_pcheck_obb":
        lv = tile_level_map(ar, cam)
        highest = np.asarray(scene["highest_levels"], F).reshape(-1)
        tmin, tb = lv["tile_min"], lv["tile_blend"]
        with np.errstate(all="ignore"):
            level_ok = lambda g, tile: tmin[tile] < (highest[g] + F(1))
            g, tile, kept = obb_pairs(ar, pr, level_ok)
        count = np.bincount(g[kept], minlength=P)
        radii = np.where(count > 0, pr["radii"], 0).astype(np.int32)
        # level_ranges (filter :374-381): single-tile rects take the tile's level as both ends, the others start from the
        # Gaussian's own highest level (lowest) and 0 (highest)
        lo = np.full(P, np.inf)
        hi = np.zeros(P)
        anyb = np.zeros(P, bool)
        np.minimum.at(lo, g[kept], tmin[tile[kept]].astype(np.float64))
        np.maximum.at(hi, g[kept], tmin[tile[kept]].astype(np.float64))
        np.logical_or.at(anyb, g[kept], tb[tile[kept]])
        multi = pr["tnum"] > 1
        lo = np.where(multi, np.minimum(lo, highest.astype(np.float64)), lo)
        single_kept = kept & ~multi[g]
        hi[g[single_kept]] = tmin[tile[single_kept]]                # one-tile rect: highest_level_used = level, no max with 0
        seen = count > 0
        lr = np.zeros((P, 2), np.int32)
        lr[seen, 0] = np.trunc(lo[seen]).astype(np.int32)
        hi_i = np.trunc(hi).astype(np.int32)
        hi_i = np.where(anyb, np.minimum(hi_i + 1, FOV_NUM - 1), hi_i)
        lr[seen, 1] = hi_i[seen]
        point_list, ranges, keys = sort_instances(pr, g, tile, kept, T)
        # compute_fov_colors
        rest = np.asarray(scene["shs"], F)
        dcs = np.asarray(scene["shs_dcs"], F)
        vis = np.nonzero(radii > 0)[0]
        d = _view_dirs(means[vis], campos)
        res = _sh_terms(ar, deg, rest[vis], d, 0)(np.zeros((vis.size, 3), F)) + lit(0.5)
        colours = np.full((P, FOV_NUM, 3), np.nan, F)
        for layer in range(FOV_NUM):
            use = (lr[vis, 0] <= layer) & (layer <= lr[vis, 1])
            col = np.fmax(lit(0.28209479177387814) * dcs[vis, layer] + res, F(0))
            colours[vis[use], layer] = col[use]
        opac4 = np.asarray(scene["opacities"], F).reshape(P, FOV_NUM)
        out.update(lv)
        out.update(level_ranges=lr, fov_colors=colours)
        out.update(blend_fov(ar, pr, cam, lv, point_list, ranges, opac4, colours, highest))
    else:
        g, tile, kept = obb_pairs(ar, pr)
        count = np.bincount(g[kept], minlength=P)
        radii = np.where(count > 0, pr["radii"], 0).astype(np.int32)
        point_list, ranges, keys = sort_instances(pr, g, tile, kept, T)
        sh = np.asarray(scene["shs"], F)
        vis = np.nonzero(pr["radii"] > 0)[0]                       # colours are evaluated in preprocess, before the box test
        d = _view_dirs(means[vis], campos)
        res = _sh_terms(ar, deg, sh[vis], d, 1)(lit(0.28209479177387814) * sh[vis, 0]) + lit(0.5)
        rgb = np.zeros((P, 3), F)
        rgb[vis] = np.fmax(res, F(0))
        opac = np.asarray(scene["opacities"], F).reshape(-1)
        out.update(blend_plain(ar, pr, cam, point_list, ranges, opac, rgb, stats=variant == "pcheck_obb_sum"))
        out["rgb"] = rgb
    out.update(radii=radii, tiles_touched=count.astype(np.uint32), point_list=point_list, ranges=ranges, keys=keys,
               num_rendered=int(point_list.size), means2D=pr["pix"], conic=pr["conic"], depths=pr["depth"],
               tiles_rect=pr["tnum"].astype(np.uint32))
    return out
