"""Pin the CPU oracle before trusting it (CPU-only).

The reference's rasterizer is CUDA-only and ships no tests, so the oracle is pinned
against (a) golden vectors produced by the reference's CPU-importable pieces
(tests/golden/make_golden.py part A), (b) textbook identities, (c) central finite
differences of its own double-precision forward for the backward pass, and (d) frozen
copies of its own outputs (regression).
"""
import math
import os

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN, cam_dict, scene_dict, small_camera, small_case, syn
from oracle import oracle as orc

VARIANTS = ("original", "pcheck_obb_sum", "pcheck_obb", "fov_pcheck_obb")


def _g(name):
    return np.load(os.path.join(GOLDEN, name))


# ---------- (a) reference-derived golden vectors ----------
def test_sh_colour_matches_reference_eval_sh():
    """oracle SH polynomial == utils/sh_utils.eval_sh(deg, sh^T, dir) + 0.5 (reference CPU code)."""
    g = _g("ref_sh.npz")
    cam = dict(image_width=16, image_height=16, tanfovx=1.0, tanfovy=1.0, bg=np.zeros(3), viewmatrix=np.eye(4),
               projmatrix=np.eye(4), campos=g["campos"], sh_degree=3)
    for deg in range(4):
        cam["sh_degree"] = deg
        scene = dict(means3D=g["pos"], opacities=np.ones((len(g["pos"]), 1)), shs=g["sh"])
        got = orc.sh_colors(scene, cam)
        np.testing.assert_allclose(got, g[f"rgb_deg{deg}"], rtol=0, atol=2e-5)
        # rest-only variant (foveated renderer): same polynomial without the DC term
        scene_r = dict(scene, shs=g["sh"][:, 1:, :])
        got_r = orc.sh_colors(scene_r, cam, rest=True)
        np.testing.assert_allclose(got_r + 0.28209479177387814 * g["sh"][:, 0, :], g[f"rgb_deg{deg}"], atol=2e-5)


def test_camera_matrices_match_reference_graphics_utils():
    g = _g("ref_camera.npz")
    for i in range(3):
        fovx, fovy = g[f"fov{i}"]
        cam = syn.MiniCam(g[f"R{i}"], g[f"T{i}"], fovx, fovy, 64, 48)
        np.testing.assert_allclose(cam.world_view_transform.numpy(), g[f"wvt{i}"], atol=1e-6)
        np.testing.assert_allclose(cam.projection_matrix.numpy(), g[f"proj{i}"], atol=1e-6)
        np.testing.assert_allclose(cam.full_proj_transform.numpy(), g[f"full{i}"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(cam.camera_center.numpy(), g[f"center{i}"], atol=1e-5)


def _ps2level(ps):
    step = (3.4641016151377544 - 1.0) / 3.0
    lv = np.where(ps <= 1, 0.0, (np.sqrt(np.maximum(ps, 1e-30)) - 1) / step)
    return np.minimum(lv, 3.9)


def test_tile_levels_match_odak_pooling_map():
    """tile level == ps2level(odak make_pooling_size_map_pixels sampled at the tile centre)."""
    g = _g("ref_pooling.npz")
    for ci in range(4):
        W, H, gx, gy, alpha = g[f"case{ci}"]
        cam = dict(image_width=int(W), image_height=int(H), tanfovx=1.0, tanfovy=1.0, bg=np.zeros(3),
                   viewmatrix=np.eye(4), projmatrix=np.eye(4), campos=np.zeros(3), sh_degree=3, gaze=(gx, gy),
                   alpha=alpha)
        lv = orc.tile_levels(cam)["tile_levels"]
        inside = g[f"inside{ci}"].reshape(-1)
        want = _ps2level(g[f"ps{ci}"].reshape(-1))
        np.testing.assert_allclose(lv[inside], want[inside], atol=3e-3)
        # double build agrees with the float build (no precision cliff in the level map)
        lv64 = orc.tile_levels(cam, dtype=np.float64)["tile_levels"]
        np.testing.assert_allclose(lv, lv64, atol=2e-4)


def test_tile_levels_match_odak_at_the_bench_gazes():
    """The nine gazes of the FPS protocol (render_compose_gazes_fps.py:26) and two off-screen ones at 1920x1080: the
    oracle's level of EVERY tile against odak's pooling-size map (tests/golden/make_golden_r3.py)."""
    g = _g("ref_pooling_gazes.npz")
    W, H = [int(x) for x in g["size"]]
    inside = g["inside"].reshape(-1)
    for gi, gaze in enumerate(g["gazes"]):
        cam = dict(image_width=W, image_height=H, tanfovx=1.0, tanfovy=1.0, bg=np.zeros(3), viewmatrix=np.eye(4), projmatrix=np.eye(4),
                   campos=np.zeros(3), sh_degree=3, gaze=(float(gaze[0]), float(gaze[1])), alpha=float(g["alpha"]))
        lv = orc.tile_levels(cam)["tile_levels"]
        want = _ps2level(g[f"ps{gi}"].astype(np.float64).reshape(-1))
        # (odak samples linspace(-0.5, 0.5, W), the rasterizer (x + 8) / W: a few 1e-3 of a level apart)
        np.testing.assert_allclose(lv[inside], want[inside], atol=4e-3, err_msg=f"gaze {gaze}")
        assert len(np.unique(np.floor(lv[inside]))) >= (4 if 0 <= gaze[0] <= 1 and 0 <= gaze[1] <= 1 else 1)


def test_sh_colour_on_the_axes_and_at_the_bench_camera():
    """eval_sh for view directions on / a hair off the coordinate axes (where degree-2 and degree-3 terms cancel or vanish)
    and for the bench camera's real view directions."""
    g = _g("ref_sh_axes.npz")
    cam = dict(image_width=16, image_height=16, tanfovx=1.0, tanfovy=1.0, bg=np.zeros(3), viewmatrix=np.eye(4),
               projmatrix=np.eye(4), campos=g["campos"], sh_degree=3)
    n_axes = int(g["n_axes"])
    for deg in range(4):
        cam["sh_degree"] = deg
        scene = dict(means3D=g["pos"], opacities=np.ones((len(g["pos"]), 1)), shs=g["sh"])
        got = orc.sh_colors(scene, cam)
        np.testing.assert_allclose(got[:n_axes], g[f"rgb_deg{deg}"][:n_axes], rtol=0, atol=3e-5, err_msg=f"degree {deg}: axis directions")
        np.testing.assert_allclose(got[n_axes:], g[f"rgb_deg{deg}"][n_axes:], rtol=0, atol=3e-5, err_msg=f"degree {deg}: bench directions")
        got_r = orc.sh_colors(dict(scene, shs=g["sh"][:, 1:, :]), cam, rest=True)
        np.testing.assert_allclose(got_r + 0.28209479177387814 * g["sh"][:, 0, :], g[f"rgb_deg{deg}"], atol=3e-5)


def test_ring_cameras_match_reference_graphics_utils():
    """The eight cameras of the bench's ring (config 5: one per GPU) against getWorld2View2 / getProjectionMatrix."""
    g = _g("ref_camera_ring.npz")
    for i in range(8):
        cam = syn.camera_ring(i, 8)
        np.testing.assert_allclose([cam.FoVx, cam.FoVy], g[f"fov{i}"], rtol=1e-12)
        np.testing.assert_allclose(cam.world_view_transform.numpy(), g[f"wvt{i}"], atol=1e-6)
        np.testing.assert_allclose(cam.projection_matrix.numpy(), g[f"proj{i}"], atol=1e-6)
        np.testing.assert_allclose(cam.full_proj_transform.numpy(), g[f"full{i}"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(cam.camera_center.numpy(), g[f"center{i}"], atol=1e-5)


# ---------- (b) identities ----------
def test_cov3d_is_R_S2_Rt():
    from scipy.spatial.transform import Rotation
    scene, cam = small_case("original", P=200)
    q = scene["rotations"].astype(np.float64)
    q /= np.linalg.norm(q, axis=1, keepdims=True)  # the kernel does NOT renormalise (forward.cu:127)
    scene = dict(scene, rotations=q)
    o = orc.forward("original", scene, cam, dtype=np.float64)
    Rm = Rotation.from_quat(q[:, [1, 2, 3, 0]]).as_matrix()  # scipy is (x,y,z,w); ours (w,x,y,z)
    S2 = scene["scales"].astype(np.float64) ** 2
    Sigma = np.einsum("nij,nj,nkj->nik", Rm, S2, Rm)
    vis = o["radii"] > 0
    want = np.stack([Sigma[:, 0, 0], Sigma[:, 0, 1], Sigma[:, 0, 2], Sigma[:, 1, 1], Sigma[:, 1, 2], Sigma[:, 2, 2]], 1)
    np.testing.assert_allclose(o["cov3D"][vis], want[vis], rtol=1e-9, atol=1e-12)


def test_projection_matches_pinhole():
    """means2D from the oracle == fx*x/z + (W-1)/2 pinhole projection for an identity camera."""
    cloud = syn.scene_1k(P=300, seed=5)
    cam = syn.camera_1k(128, 96, 60.0)
    # camera_1k uses the same FoV on both axes; recompute focal lengths from tan
    sc, cd = scene_dict(cloud, "original"), cam_dict(cam)
    o = orc.forward("original", sc, cd, dtype=np.float64)
    vis = o["radii"] > 0
    xyz = sc["means3D"].astype(np.float64)
    fx = 128 / (2 * cd["tanfovx"])
    fy = 96 / (2 * cd["tanfovy"])
    u = fx * xyz[:, 0] / xyz[:, 2] + (128 - 1) / 2
    v = fy * xyz[:, 1] / xyz[:, 2] + (96 - 1) / 2
    np.testing.assert_allclose(o["means2D"][vis, 0], u[vis], atol=1e-4)
    np.testing.assert_allclose(o["means2D"][vis, 1], v[vis], atol=1e-4)
    np.testing.assert_allclose(o["depths"][vis], xyz[vis, 2], atol=1e-6)


def test_sorted_list_is_stable_by_tile_depth_index():
    scene, cam = small_case("pcheck_obb_sum")
    o = orc.forward("pcheck_obb_sum", scene, cam)
    keys = o["keys"]
    assert np.all(keys[1:] >= keys[:-1])
    same = keys[1:] == keys[:-1]
    pl = o["point_list"].astype(np.int64)
    assert np.all(pl[1:][same] > pl[:-1][same])  # ties resolved by ascending Gaussian index
    tiles = (keys >> np.uint64(32)).astype(np.int64)
    for t in np.unique(tiles):
        lo, hi = o["ranges"][t]
        assert np.all(tiles[lo:hi] == t) and (hi - lo) == np.sum(tiles == t)
    assert int(o["tiles_touched"].sum()) == o["num_rendered"]


@pytest.mark.parametrize("variant", VARIANTS)
def test_float_and_double_builds_agree(variant):
    scene, cam = small_case(variant)
    o32 = orc.forward(variant, scene, cam, dtype=np.float32)
    o64 = orc.forward(variant, scene, cam, dtype=np.float64)
    diff = np.abs(o32["color"] - o64["color"])
    # discrete decisions (radius ceil, tile rects, 1/255 and 1e-4 thresholds) may flip for a
    # handful of pixels between precisions; everything else agrees to fp32 round-off
    assert np.mean(diff > 1e-4) < 2e-3
    assert np.median(diff) < 1e-6


# ---------- (c) backward vs finite differences (double build) ----------
def _loss(variant, scene, cam, wpix):
    o = orc.forward(variant, scene, cam, dtype=np.float64)
    return float((o["color"] * wpix).sum()), o


@pytest.mark.parametrize("variant", ("original", "pcheck_obb_sum"))
def test_backward_matches_finite_differences(variant):
    """d(sum(color*w))/d(param) from orc_backward == central differences of orc_forward (fp64)."""
    cloud = syn.scene_1k(P=24, seed=11)
    cloud._scaling += 0.8
    cam = syn.camera_1k(48, 32, 60.0)
    scene = {k: v.astype(np.float64) for k, v in scene_dict(cloud, variant).items()}
    cd = cam_dict(cam, bg=(0.3, 0.1, 0.2))
    rng = np.random.default_rng(0)
    wpix = rng.normal(size=(3, 32, 48))
    base, o = _loss(variant, scene, cd, wpix)
    g = orc.backward(variant, scene, cd, o, wpix, dtype=np.float64)
    checks = [("means3D", "dL_dmean3D"), ("scales", "dL_dscale"), ("rotations", "dL_drot"),
              ("opacities", "dL_dopacity"), ("shs", "dL_dsh")]
    n_checked = 0
    for pname, gname in checks:
        arr = scene[pname]
        ga = g[gname].reshape(arr.shape)
        flat_idx = rng.choice(arr.size, size=min(40, arr.size), replace=False)
        for fi in flat_idx:
            idx = np.unravel_index(fi, arr.shape)
            h = 1e-6 * max(1.0, abs(arr[idx]))
            old = arr[idx]
            arr[idx] = old + h
            lp, op = _loss(variant, scene, cd, wpix)
            arr[idx] = old - h
            lm, om = _loss(variant, scene, cd, wpix)
            arr[idx] = old
            # skip probes that cross a discrete decision (visible set / list / contributor changes)
            if not (np.array_equal(op["n_contrib"], om["n_contrib"]) and np.array_equal(op["point_list"], om["point_list"])
                    and np.array_equal(op["point_list"], o["point_list"]) and np.array_equal(op["n_contrib"], o["n_contrib"])):
                continue
            fd = (lp - lm) / (2 * h)
            assert abs(fd - ga[idx]) <= 2e-5 * max(1.0, abs(fd), abs(ga[idx])), (pname, idx, fd, ga[idx])
            n_checked += 1
    assert n_checked > 100


def test_backward_means2D_gradient_is_screen_space_gradient():
    """dL_dmean2D (the viewspace_points grad) equals d loss / d(ndc) * 0.5*W: check by moving a
    Gaussian along camera x and comparing with the 3D-mean gradient projected through J."""
    cloud = syn.scene_1k(P=8, seed=21)
    cam = syn.camera_1k(32, 32, 60.0)
    scene = {k: v.astype(np.float64) for k, v in scene_dict(cloud, "original").items()}
    cd = cam_dict(cam)
    rng = np.random.default_rng(1)
    wpix = rng.normal(size=(3, 32, 32))
    _, o = _loss("original", scene, cd, wpix)
    g = orc.backward("original", scene, cd, o, wpix, dtype=np.float64)
    assert g["dL_dmean2D"].shape == (8, 3) and np.all(g["dL_dmean2D"][:, 2] == 0)
    assert np.any(g["dL_dmean2D"][:, :2] != 0)


# ---------- (d) frozen fixtures ----------
@pytest.mark.parametrize("variant", VARIANTS)
def test_frozen_fixture(variant):
    g = _g(f"oracle_{variant}.npz")
    scene, cam = small_case(variant)
    o = orc.forward(variant, scene, cam)
    assert o["num_rendered"] == int(g["num_rendered"])
    np.testing.assert_array_equal(o["radii"], g["radii"])
    np.testing.assert_array_equal(o["point_list"], g["point_list"])
    np.testing.assert_array_equal(o["ranges"], g["ranges"])
    np.testing.assert_allclose(o["color"], g["color"], atol=1e-6)
    if variant == "pcheck_obb_sum":
        np.testing.assert_array_equal(o["gaussians_count"], g["gaussians_count"])
    if variant in ("original", "pcheck_obb_sum"):
        gr = orc.backward(variant, scene, cam, o, g["dL_dpix"])
        for k, v in gr.items():
            np.testing.assert_allclose(v, g["g_" + k], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("variant", ("pcheck_obb_max", "pcheck_obb_loss_weighted_max_count"))
def test_pruning_metric_variants(variant):
    """Same image / lists as pcheck_obb_sum, different per-Gaussian statistics; frozen fixture + invariants."""
    g = _g(f"oracle_{variant}.npz")
    scene, cam = small_case(variant)
    o = orc.forward(variant, scene, cam)
    ref = orc.forward("pcheck_obb_sum", {k: v for k, v in scene.items() if k != "loss_map"}, cam)
    np.testing.assert_array_equal(o["color"], ref["color"])
    np.testing.assert_array_equal(o["point_list"], ref["point_list"])
    np.testing.assert_array_equal(o["gaussians_count"], g["gaussians_count"])
    np.testing.assert_allclose(o["contributions"], g["contributions"], rtol=1e-6, atol=1e-7)
    if variant == "pcheck_obb_max":
        # max over pixels of alpha*T is bounded by 0.99 and only set where something contributed
        assert o["contributions"].max() <= 0.99 + 1e-6 and np.all(o["gaussians_count"][o["contributions"] > 0] > 0)
    else:
        # every pixel credits its loss to exactly one Gaussian; fetch counts are those of pcheck_obb_sum
        np.testing.assert_allclose(o["contributions"].sum(), scene["loss_map"][0].sum(), rtol=1e-5)
        np.testing.assert_array_equal(o["gaussians_count"], ref["gaussians_count"])


def test_edge_cases():
    # empty cloud
    scene, cam = small_case("original", P=50)
    empty = {k: v[:0] for k, v in scene.items()}
    o = orc.forward("original", empty, cam)
    assert o["num_rendered"] == 0 and np.all(o["color"] == 0)
    # everything behind the camera -> background only
    behind = dict(scene)
    behind["means3D"] = scene["means3D"].copy()
    behind["means3D"][:, 2] = -5.0
    o = orc.forward("pcheck_obb_sum", behind, cam)
    assert o["num_rendered"] == 0
    for ch in range(3):
        np.testing.assert_allclose(o["color"][ch], cam["bg"][ch])
    assert not orc.mark_visible(behind, cam).any()
    assert orc.mark_visible(scene, cam).sum() > 0


def test_shared_model_foveated_baseline_pins():
    """The SMFR restatement (…_naive_pcheck_obb) against the two variants it is made of: with every Gaussian present at
    all levels (highest_levels == 3) its instance lists are pcheck_obb's and every single-level tile is blended exactly
    like pcheck_obb's (same formula, NV forward.cu:482-580 vs RP forward.cu:243-384); with the foveated model's
    highest_levels its radii and lists are RF's (same filter, NV rasterizer_impl.cu:264-357 vs RF :264-383)."""
    import numpy as np
    from tests.helpers import small_case
    scene_f, cam = small_case("fov_pcheck_obb", P=3000, seed=5, gaze=(0.3, 0.6), width=320, height=208)
    plain, _ = small_case("pcheck_obb", P=3000, seed=5, width=320, height=208)
    sm = dict(plain, highest_levels=np.full((3000, 1), 3.0, np.float32))
    a, b = orc.forward("naive_pcheck_obb", sm, cam), orc.forward("pcheck_obb", plain, cam)
    assert a["num_rendered"] == b["num_rendered"] > 10000
    np.testing.assert_array_equal(a["point_list"], b["point_list"])
    two = a["tile_blend"].reshape(13, 20) != 0
    assert 20 < two.sum() < 200
    single_px = ~np.repeat(np.repeat(two, 16, 0), 16, 1)[:208, :320]
    np.testing.assert_array_equal(a["color"][:, single_px], b["color"][:, single_px])
    assert 1e-5 < np.abs(a["color"] - b["color"])[:, ~single_px].max() < 0.2  # two-level tiles mix two states
    c = orc.forward("naive_pcheck_obb", dict(plain, highest_levels=scene_f["highest_levels"]), cam)
    f = orc.forward("fov_pcheck_obb", scene_f, cam)
    np.testing.assert_array_equal(c["radii"], f["radii"])
    np.testing.assert_array_equal(c["point_list"], f["point_list"])
    np.testing.assert_array_equal(c["ranges"], f["ranges"])


def test_multi_model_baseline_partition_of_unity():
    """The MMFR restatement (…_mmfr_pcheck_obb): the level bands (cur_level - 0.5, cur_level + 1) cover every tile, a
    single-level tile is rendered by exactly one level, and the smoothstep weights of a two-level tile's two levels add
    up to one -- so the four level renders of ONE model must add up to that model's plain pcheck_obb image."""
    import numpy as np
    from tests.helpers import small_case
    for gaze in ((0.3, 0.6), (0.95, 0.05)):
        plain, cam = small_case("pcheck_obb", P=3000, seed=5, gaze=gaze, width=320, height=208)
        total, counts = 0, []
        for level in range(4):
            o = orc.forward("mmfr_pcheck_obb", plain, dict(cam, cur_level=float(level)))
            total = total + o["color"]
            counts.append(o["num_rendered"])
        want = orc.forward("pcheck_obb", plain, cam)
        assert sum(counts) >= want["num_rendered"] and min(counts) > 0
        np.testing.assert_allclose(total, want["color"], rtol=0, atol=3e-7)


def test_reference_arithmetic_noise_of_the_gradients():
    """What "1e-4 relative" can mean for the gradients: the fp32 restatement of the reference against the same arithmetic
    in double, on the SAME forward state (lists, n_contrib, final_T of the fp32 forward, so that no discrete decision
    differs). The backward pass recovers T by repeated division and forms dL/dalpha from colour differences
    (R0 backward.cu:503-521); rows whose terms cancel carry the rounding of their summands. The spread measured here --
    the reference's own fp32 arithmetic against exact arithmetic -- is what the row budgets of tests/checks.py
    (GRAD_OUTLIERS / GRAD_GROSS) are sized for; it is recorded next to the HIP-vs-oracle numbers."""
    from tests.checks import grad_stats
    from tests import parity_report
    variant = "pcheck_obb_sum"
    scene = scene_dict(syn.scene_1k(P=6000, seed=3), variant)  # (translucent: most Gaussians reach a pixel)
    cam = cam_dict(small_camera(320, 200))
    o32 = orc.forward(variant, scene, cam)
    wpix = np.random.default_rng(5).normal(size=o32["color"].shape).astype(np.float32)
    g32 = orc.backward(variant, scene, cam, o32, wpix)
    # the double build fed with the fp32 forward's per-Gaussian state and lists
    o64 = {k: (v.astype(np.float64) if isinstance(v, np.ndarray) and v.dtype == np.float32 else v) for k, v in o32.items()}
    g64 = orc.backward(variant, {k: v.astype(np.float64) for k, v in scene.items()}, cam, o64, wpix.astype(np.float64), dtype=np.float64)
    worst = 0.0
    for k in ("dL_dmean2D", "dL_dopacity", "dL_dcolor", "dL_dmean3D", "dL_dsh", "dL_dscale", "dL_drot"):
        st = grad_stats(g32[k], g64[k])
        parity_report.record("oracle_noise", f"fp32 vs fp64 oracle backward {k}", **st)
        worst = max(worst, st["frac_bad"])
        assert st["one_minus_cosine"] < 1e-7 and st["row_rel_p50"] < 1e-5, (k, st)
    assert worst <= 3e-3, worst  # the fp32 reference arithmetic itself stays inside the budget the HIP path is held to


# ---------- (e) an independent derivation of the gradients: torch.autograd through a naive PyTorch-CPU rasterizer ----------
@pytest.mark.parametrize("variant", ("original", "pcheck_obb_sum"))
def test_backward_matches_autograd_of_a_torch_rasterizer(variant):
    """BASELINE config 1 (S-1k cloud, a naive PyTorch-CPU forward rasterizer: tests/torch_rasterizer.py; 128 x 128 here -- the
    P x pixels tables of the autograd graph are 4 x smaller than at 256 x 256, which only the forward is run at, for its time):
    its image equals the oracle's double-precision forward to the last digits the float-typed literals of the C source leave
    (0.3f, 1e-7f, the SH constants: 3e-8), and torch.autograd through it gives the gradients of all six parameter tensors that
    orc_backward (the line-by-line restatement of R0/cuda_rasterizer/backward.cu) computes -- a second, independent derivation
    beside the finite-difference probes above. pcheck_obb_sum: the RS blend rule (power < -4.5 skipped,
    RS forward.cu:376-380, backward.cu:495) on R0's tile rectangles: a cloud of small splats (one tile each, where RS applies no
    box test, RS rasterizer_impl.cu:99-102), so that both rasterizers see the same lists."""
    import time
    from tests.torch_rasterizer import rasterize
    cloud = syn.scene_1k(P=1000, seed=0)
    if variant == "pcheck_obb_sum":
        cloud._scaling -= 1.6  # sub-tile splats: rect of one tile, no OBB test in RS
    cam = syn.camera_1k(128, 128)
    scene = {k: v.astype(np.float64) for k, v in scene_dict(cloud, variant).items()}
    cd = cam_dict(cam, bg=(0.2, 0.4, 0.1))
    o = orc.forward(variant, scene, cd, dtype=np.float64)
    if variant == "pcheck_obb_sum":
        multi = (o["tiles_rect"] > 1) & (o["radii"] > 0)  # tiles_rect: tiles of the splat's rectangle before the box test
        keep = ~multi  # the few splats that still straddle a tile boundary are left out (RS would box-test them)
        scene = {k: v[keep] for k, v in scene.items()}
        o = orc.forward(variant, scene, cd, dtype=np.float64)
    assert (o["radii"] > 0).sum() > 400
    t_fwd = None
    if variant == "original":  # the baseline's forward time at config 1's size (no autograd graph)
        cd256 = cam_dict(syn.camera_1k(256, 256), bg=(0.2, 0.4, 0.1))
        with torch.no_grad():
            t0 = time.perf_counter()
            img256 = rasterize(*(torch.tensor(scene[k], dtype=torch.float64) for k in ("means3D", "scales", "rotations", "opacities", "shs")), cd256)
            t_fwd = time.perf_counter() - t0
        o256 = orc.forward(variant, scene, cd256, dtype=np.float64)
        assert np.abs(img256.numpy() - o256["color"]).max() < 1e-6
    params = {k: torch.tensor(scene[k], dtype=torch.float64, requires_grad=True) for k in ("means3D", "scales", "rotations", "opacities", "shs")}
    img = rasterize(params["means3D"], params["scales"], params["rotations"], params["opacities"], params["shs"], cd,
                    cutoff=variant != "original")
    d = np.abs(img.detach().numpy() - o["color"])
    assert d.max() < 1e-6, d.max()
    rng = np.random.default_rng(3)
    wpix = rng.normal(size=o["color"].shape)
    (img * torch.tensor(wpix)).sum().backward()
    g = orc.backward(variant, scene, cd, o, wpix, dtype=np.float64)
    from tests import parity_report
    for pname, gname in (("means3D", "dL_dmean3D"), ("scales", "dL_dscale"), ("rotations", "dL_drot"), ("opacities", "dL_dopacity"), ("shs", "dL_dsh")):
        want, got = g[gname].reshape(scene[pname].shape), params[pname].grad.numpy()
        scale = np.abs(want).max()
        err = np.abs(got - want).max()
        parity_report.record("autograd", f"{variant} {gname}: oracle backward vs torch.autograd", max_abs=float(err), ref_max=float(scale), n=int(want.size))
        assert scale > 0 and err <= 1e-5 * max(1.0, scale), (pname, err, scale)  # (measured: 5e-8 original, 4e-6 on the sub-tile splats, whose covariance is mostly the 0.3f dilation)
    if t_fwd is not None:
        parity_report.record("cpu_torch_baseline", f"{variant}: naive PyTorch-CPU forward, S-1k @ 256x256", seconds=float(t_fwd), threads=int(torch.get_num_threads()))
