"""GPU parity: the HIP library (through its C ABI) against the CPU oracle on the same seeded inputs.

Tolerances (fp32 path; BASELINE north_star: per-pixel match within 1e-4):
  * integer / index outputs (radii, instance count, per-tile ranges, sorted id lists): bit-exact;
  * images and gradients: tests/checks.py (every comparison's measured maximum and outlier fraction is kept in
    tests/parity_report_gpu.json).
"""
import math
import os

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN, cam_dict, scene_dict, small_camera, small_case, small_cloud, syn
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

VARIANTS = ("original", "pcheck_obb_sum", "pcheck_obb", "fov_pcheck_obb")


def _need_gpu():
    if not torch.cuda.is_available():  # only reached by an explicit -m gpu run (tests/conftest.py skips otherwise)
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")


from tests.checks import check_grad, check_image  # noqa: E402  (record what they measure: tests/parity_report_gpu.json)


@pytest.mark.parametrize("variant", VARIANTS)
def test_forward_matches_oracle(variant):
    _need_gpu()
    from tests.gpu_helpers import hip_forward
    scene, cam = small_case(variant)
    want = orc.forward(variant, scene, cam)
    got = hip_forward(variant, scene, cam)
    assert got["num_rendered"] == want["num_rendered"]
    np.testing.assert_array_equal(got["radii"], want["radii"])
    np.testing.assert_array_equal(got["ranges"], want["ranges"])
    np.testing.assert_array_equal(got["point_list"], want["point_list"])
    check_image(got["color"], want["color"])
    if variant in ("original", "pcheck_obb_sum"):
        assert np.mean(got["n_contrib"] != want["n_contrib"]) <= 1e-3
        same = got["n_contrib"] == want["n_contrib"]
        np.testing.assert_allclose(got["final_T"][same], want["final_T"][same], rtol=1e-4, atol=1e-7)
    if variant == "pcheck_obb_sum":
        assert np.mean(got["gaussians_count"] != want["gaussians_count"]) <= 1e-3
        check_grad(got["contributions"], want["contributions"], "contributions")
    if variant == "fov_pcheck_obb":
        np.testing.assert_allclose(got["tile_levels"], want["tile_levels"], atol=2e-5)
        np.testing.assert_allclose(got["tile_min"], want["tile_min"], atol=2e-5)
        np.testing.assert_array_equal(got["tile_blend"], want["tile_blend"])


@pytest.mark.parametrize("variant,gaze", [("pcheck_obb", (0.5, 0.5)), ("fov_pcheck_obb", (0.08, 0.9)),
                                          ("fov_pcheck_obb", (0.7, 0.3)), ("fov_pcheck_obb", (1.4, -0.2))])
def test_clipped_walk_rectangles(variant, gaze):
    """The binning kernels walk getRect() clipped to the OBB's axis-aligned box and to the box of the tiles
    whose level passes (common.h walk_rect); the instance lists must stay the reference's bit for bit.
    Needle-shaped, diagonal and frame-filling splats on a 1280x720 frame, gaze near a corner / off screen."""
    _need_gpu()
    from tests.gpu_helpers import hip_forward
    cloud = small_cloud(6000, seed=11, big_fraction=0.3)
    g = torch.Generator().manual_seed(5)
    needle = torch.randperm(6000, generator=g)[:2500]
    cloud._scaling[needle, 0] += 2.5   # up to ~100:1 aspect ratios
    cloud._scaling[needle, 1] -= 1.5
    cloud._scaling[needle[:40]] += 2.0  # a few cover most of the frame
    cam = small_camera(1280, 720)
    fov = syn.foveation_layers(cloud, seed=12) if variant == "fov_pcheck_obb" else None
    scene = scene_dict(cloud, variant, fov)
    cd = cam_dict(cam, bg=(0.0, 0.1, 0.0), gaze=gaze)
    want = orc.forward(variant, scene, cd)
    got = hip_forward(variant, scene, cd)
    assert want["num_rendered"] > 50000
    assert got["num_rendered"] == want["num_rendered"]
    np.testing.assert_array_equal(got["radii"], want["radii"])
    np.testing.assert_array_equal(got["ranges"], want["ranges"])
    np.testing.assert_array_equal(got["point_list"], want["point_list"])
    check_image(got["color"], want["color"])


@pytest.mark.parametrize("variant", VARIANTS)
def test_forward_matches_frozen_fixture(variant):
    _need_gpu()
    from tests.gpu_helpers import hip_forward
    g = np.load(os.path.join(GOLDEN, f"oracle_{variant}.npz"))
    scene, cam = small_case(variant)
    got = hip_forward(variant, scene, cam)
    assert got["num_rendered"] == int(g["num_rendered"])
    np.testing.assert_array_equal(got["radii"], g["radii"])
    np.testing.assert_array_equal(got["point_list"], g["point_list"])
    check_image(got["color"], g["color"])


@pytest.mark.parametrize("variant", ("pcheck_obb_max", "pcheck_obb_loss_weighted_max_count"))
def test_pruning_metric_variants(variant):
    """the two pruning-metric flavours of the training rasterizer."""
    _need_gpu()
    from tests.gpu_helpers import hip_backward, hip_forward
    scene, cam = small_case(variant)
    want = orc.forward(variant, scene, cam)
    got = hip_forward(variant, scene, cam)
    assert got["num_rendered"] == want["num_rendered"]
    np.testing.assert_array_equal(got["point_list"], want["point_list"])
    check_image(got["color"], want["color"])
    assert np.mean(got["gaussians_count"] != want["gaussians_count"]) <= 2e-3
    if variant == "pcheck_obb_max":
        check_grad(got["contributions"], want["contributions"], "contributions(max)", rtol=1e-5)
    else:
        # a pixel whose two best contributions are within an ulp may credit the other Gaussian
        np.testing.assert_allclose(got["contributions"].sum(), want["contributions"].sum(), rtol=1e-5)
        assert np.mean(np.abs(got["contributions"] - want["contributions"]) > 1e-3) <= 2e-3
    rng = np.random.default_rng(7)
    dpix = rng.normal(size=want["color"].shape).astype(np.float32)
    wb = orc.backward(variant, scene, cam, want, dpix)
    gb = hip_backward(variant, got, dpix)
    for k in ("dL_dmean3D", "dL_dsh", "dL_dscale", "dL_drot", "dL_dopacity"):
        check_grad(gb[k].reshape(wb[k].shape), wb[k], k)


def test_max_variant_blends_the_pairs_the_backward_pass_takes():
    """pcheck_obb_max counts pixels BEFORE the alpha test, so its forward blend kept the reference's own form of that test
    while k_render_bwd (shared by the training variants) decides by the threshold on q. Now the _max blend applies the same threshold
    (a second one beside the support's): its image, final_T and n_contrib are the training variant's BIT FOR BIT on a cloud whose
    opacities crowd the 1/255 line, and its gradients those of the training variant. (A guard on the structure -- one test, one blend
    function, entries that pass nowhere -- more than on the rounding: a flip needs a pair within an ulp of the line, a few per whole
    1080p frame, and the previous library happens to pass on this scene too.)"""
    _need_gpu()
    from tests.gpu_helpers import hip_backward, hip_forward
    scene, cam = small_case("pcheck_obb_max", P=30000, seed=21, width=960, height=544)
    rng = np.random.default_rng(3)
    op = np.asarray(scene["opacities"], np.float32).copy()
    low = rng.random(op.shape[0]) < 0.6
    op[low, 0] = rng.uniform(0.5 / 255.0, 6.0 / 255.0, int(low.sum())).astype(np.float32)   # alpha = o e^-q crosses 1/255 inside the splat
    op[rng.random(op.shape[0]) < 0.02, 0] = -0.1                                            # ... and a few that pass nowhere
    scene = dict(scene, opacities=op)
    a = hip_forward("pcheck_obb_max", scene, cam)
    b = hip_forward("pcheck_obb_sum", scene, cam)
    assert a["num_rendered"] == b["num_rendered"] > 100_000
    np.testing.assert_array_equal(a["point_list"], b["point_list"])
    np.testing.assert_array_equal(a["color"], b["color"])
    np.testing.assert_array_equal(a["final_T"], b["final_T"])
    np.testing.assert_array_equal(a["n_contrib"], b["n_contrib"])
    want = orc.forward("pcheck_obb_max", scene, cam)
    check_image(a["color"], want["color"])
    assert np.mean(a["gaussians_count"] != want["gaussians_count"]) <= 2e-3   # the support count is untouched by the alpha threshold
    dpix = rng.normal(size=a["color"].shape).astype(np.float32)
    ga, gb = hip_backward("pcheck_obb_max", a, dpix), hip_backward("pcheck_obb_sum", b, dpix)
    for k in ("dL_dmean3D", "dL_dsh", "dL_dscale", "dL_drot", "dL_dopacity"):
        check_grad(ga[k], gb[k], k + " (_max against _sum)")


@pytest.mark.parametrize("variant", ("original", "pcheck_obb_sum"))
def test_backward_matches_oracle(variant):
    _need_gpu()
    from tests.gpu_helpers import hip_backward, hip_forward
    scene, cam = small_case(variant)
    want_f = orc.forward(variant, scene, cam)
    rng = np.random.default_rng(7)
    dpix = rng.normal(size=want_f["color"].shape).astype(np.float32)
    want = orc.backward(variant, scene, cam, want_f, dpix)
    got_f = hip_forward(variant, scene, cam)
    got = hip_backward(variant, got_f, dpix)
    for k in ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot"):
        check_grad(got[k].reshape(want[k].shape), want[k], k)


@pytest.mark.parametrize("variant", ("original", "pcheck_obb_sum"))
def test_backward_with_precomputed_colours_and_covariances(variant):
    """colors_precomp + cov3D_precomp instead of SH coefficients, scales and rotations (the rasterizer's other input form,
    diff_gaussian_rasterization/__init__.py:60-75): the backward pass then returns dL_dcolors / dL_dcov3D (backward.cu:346-396 writes
    them either way) and zeros for the tensors of the absent inputs. Every output starts as NaN (tests/conftest.py): positions and
    opacity go out in whole lines, colours and covariances row by row over the fill, scales / rotations are the fill's alone."""
    _need_gpu()
    from tests.gpu_helpers import hip_backward, hip_forward
    scene, cam = small_case(variant)
    w0 = orc.forward(variant, scene, cam)
    pre = dict(scene)
    pre["colors_precomp"] = np.random.default_rng(3).random((scene["means3D"].shape[0], 3)).astype(np.float32)
    pre["cov3D_precomp"] = w0["cov3D"]
    for k in ("shs", "scales", "rotations"):
        pre.pop(k)
    want_f = orc.forward(variant, pre, cam)
    dpix = np.random.default_rng(8).normal(size=want_f["color"].shape).astype(np.float32)
    want = orc.backward(variant, pre, cam, want_f, dpix)
    got_f = hip_forward(variant, pre, cam)
    np.testing.assert_array_equal(got_f["point_list"], want_f["point_list"])
    got = hip_backward(variant, got_f, dpix)
    for k in ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D"):
        check_grad(got[k].reshape(want[k].shape), want[k], k + " (precomputed inputs)")
    for k in ("dL_dsh", "dL_dscale", "dL_drot"):
        assert got[k].size == 0 or not np.abs(got[k]).any(), k  # (NaN != 0: unwritten elements fail here too)
        assert np.isfinite(got[k]).all(), k


@pytest.mark.parametrize("sh_degree", (0, 1, 2))
def test_backward_at_lower_sh_degrees(sh_degree):
    """Active SH degree below the allocated one (early training, scene/gaussian_model.py oneupSHdegree): only the coefficients of
    the active degree are read and get a gradient (backward.cu:20-139), the rest of the [P,16,3] row stays zero -- the path of
    k_preprocess_bwd that moves a wave's SH rows together handles rows that are only partly used."""
    _need_gpu()
    from tests.gpu_helpers import hip_backward, hip_forward
    scene, cam = small_case("pcheck_obb_sum")
    cam = dict(cam, sh_degree=sh_degree)
    want_f = orc.forward("pcheck_obb_sum", scene, cam)
    dpix = np.random.default_rng(17 + sh_degree).normal(size=want_f["color"].shape).astype(np.float32)
    want = orc.backward("pcheck_obb_sum", scene, cam, want_f, dpix)
    got_f = hip_forward("pcheck_obb_sum", scene, cam)
    check_image(got_f["color"], want_f["color"])
    got = hip_backward("pcheck_obb_sum", got_f, dpix)
    for k in ("dL_dmean2D", "dL_dopacity", "dL_dmean3D", "dL_dsh", "dL_dscale", "dL_drot"):
        check_grad(got[k].reshape(want[k].shape), want[k], f"{k} (degree {sh_degree})")
    used = (sh_degree + 1) ** 2
    assert np.all(got["dL_dsh"].reshape(-1, 16, 3)[:, used:] == 0.0)


@pytest.mark.parametrize("variant", ("original", "fov_pcheck_obb"))
def test_1k_scene_256(variant):
    """BASELINE config 1 (S-1k, 256x256) on the GPU path."""
    _need_gpu()
    from tests.gpu_helpers import hip_forward
    cloud = syn.scene_1k()
    cam = syn.camera_1k()
    fov = syn.foveation_layers(cloud) if variant == "fov_pcheck_obb" else None
    scene, cd = scene_dict(cloud, variant, fov), cam_dict(cam, gaze=(0.25, 0.75))
    want = orc.forward(variant, scene, cd)
    got = hip_forward(variant, scene, cd)
    assert got["num_rendered"] == want["num_rendered"]
    np.testing.assert_array_equal(got["point_list"], want["point_list"])
    check_image(got["color"], want["color"])


def test_edge_cases():
    _need_gpu()
    from tests.gpu_helpers import hip_forward
    scene, cam = small_case("pcheck_obb_sum", P=64)
    # everything behind the camera: background only, zero instances
    behind = dict(scene)
    behind["means3D"] = scene["means3D"].copy()
    behind["means3D"][:, 2] = -5.0
    got = hip_forward("pcheck_obb_sum", behind, cam)
    assert got["num_rendered"] == 0 and not got["radii"].any()
    for ch in range(3):
        np.testing.assert_allclose(got["color"][ch], cam["bg"][ch])
    # empty cloud: zero image (reference returns the zero-initialised tensor)
    empty = {k: v[:0] for k, v in scene.items()}
    got = hip_forward("original", empty, cam)
    assert got["num_rendered"] == 0 and np.all(got["color"] == 0)
    # one huge splat covering every tile, and precomputed colours / covariances
    one = {k: v[:1].copy() for k, v in scene.items()}
    one["means3D"][:] = (0.0, 0.0, 4.0)
    one["scales"][:] = 3.0
    want = orc.forward("original", one, cam)
    got = hip_forward("original", one, cam)
    assert got["num_rendered"] == want["num_rendered"] == want["ranges"].shape[0]
    check_image(got["color"], want["color"])
    pre = dict(scene)
    w0 = orc.forward("original", scene, cam)
    pre["colors_precomp"] = np.random.default_rng(0).random((64, 3)).astype(np.float32)
    pre["cov3D_precomp"] = w0["cov3D"]
    pre.pop("shs"); pre.pop("scales"); pre.pop("rotations")
    want = orc.forward("original", pre, cam)
    got = hip_forward("original", pre, cam)
    np.testing.assert_array_equal(got["point_list"], want["point_list"])
    check_image(got["color"], want["color"])


@pytest.mark.parametrize("depths", ("spread", "walls"))
@pytest.mark.parametrize("P", (400, 700, 3000, 6000, 9000, 18000, 24000))
def test_tile_sort_size_classes(P, depths):
    """Every size class of the per-tile sort -- <= 512 entries (one wave), 513..2047, 2048..4095, 4096..8191, 8192..16383: the
    whole list in LDS (interpolation sort: bucket by depth, then odd-even rounds inside the buckets) -- and the lists beyond
    that, regrouped by depth into chunks first. "walls": the depths take 7 values only, thousands of entries share a
    bucket, the sort falls back to the merge passes (and the regrouping to the global-memory network) and the order is
    decided by the Gaussian index alone."""
    _need_gpu()
    from tests.gpu_helpers import hip_forward
    cloud = syn.scene_1k(P=P, seed=9)
    cloud._xyz[:, :2] *= 0.02      # pile everything onto a few tiles
    cloud._opacity -= 3.5          # faint, so nothing saturates early
    if depths == "walls":
        cloud._xyz[:, 2] = torch.round(cloud._xyz[:, 2] * 3.0) / 3.0  # identity camera: view depth = z
    cam = syn.camera_1k(64, 64)
    scene, cd = scene_dict(cloud, "original"), cam_dict(cam)
    want = orc.forward("original", scene, cd)
    assert (want["ranges"][:, 1] - want["ranges"][:, 0]).max() > 0.9 * P
    got = hip_forward("original", scene, cd)
    np.testing.assert_array_equal(got["point_list"], want["point_list"])
    check_image(got["color"], want["color"])


def test_1440p_tile_grid_keeps_tables_in_lds():
    """2560x1440 -> 14400 tiles: the largest class of tile grids whose per-workgroup histogram (56 KiB) and RF tile
    table (58 KiB) still live in LDS (one binning workgroup per CU)."""
    _need_gpu()
    from tests.gpu_helpers import hip_forward
    cloud = syn.scene_1k(P=1200, seed=21)
    cloud._scaling += 0.4
    cam = syn.camera_1k(2560, 1440, 70.0)
    fov = syn.foveation_layers(cloud, seed=22)
    scene, cd = scene_dict(cloud, "fov_pcheck_obb", fov), cam_dict(cam, gaze=(0.35, 0.55))
    want = orc.forward("fov_pcheck_obb", scene, cd)
    for packed in (False, True):
        got = hip_forward("fov_pcheck_obb", scene, cd, packed=packed)
        assert got["num_rendered"] == want["num_rendered"]
        np.testing.assert_array_equal(got["radii"], want["radii"])
        np.testing.assert_array_equal(got["ranges"], want["ranges"])
        np.testing.assert_array_equal(got["point_list"], want["point_list"])
        check_image(got["color"], want["color"])


@pytest.mark.parametrize("variant", ("pcheck_obb", "fov_pcheck_obb"))
def test_huge_tile_grid_uses_global_counter_path(variant):
    """More than 16384 tiles (4096x2160 -> 34560): the per-workgroup LDS histograms / cursors hold 16-bit counts, two tiles
    per word (up to 34816 tiles; the 8K test below is on global per-tile counters / cursors)."""
    _need_gpu()
    from tests.gpu_helpers import hip_forward
    cloud = syn.scene_1k(P=1500, seed=12)
    cloud._scaling += 0.5
    cam = syn.camera_1k(4096, 2160, 70.0)
    fov = syn.foveation_layers(cloud, seed=13) if variant == "fov_pcheck_obb" else None
    scene, cd = scene_dict(cloud, variant, fov), cam_dict(cam, gaze=(0.6, 0.4))
    want = orc.forward(variant, scene, cd)
    got = hip_forward(variant, scene, cd)
    assert got["num_rendered"] == want["num_rendered"]
    np.testing.assert_array_equal(got["ranges"], want["ranges"])
    np.testing.assert_array_equal(got["point_list"], want["point_list"])
    check_image(got["color"], want["color"])
    # a packed model on this path: the cull pass reads it, the binning kernel falls back to the ordinary tensors
    pk = hip_forward(variant, scene, cd, packed=True)
    for k in ("radii", "ranges", "point_list", "color"):
        np.testing.assert_array_equal(pk[k], got[k], err_msg="packed " + k)


def test_4k_tile_grid_with_a_large_cloud():
    """4096x2160 (34 560 tiles: 16-bit histograms, two tiles per word) with 600 000 Gaussians: the cull pass then runs its full grid of
    8192 waves, whose 32 KiB of running counts no longer fit into k_bin's LDS beside the histogram (it reads them from global
    memory), and every binning wave takes several slabs (static order: wave, wave + waves, ...). Lists and image against the oracle."""
    _need_gpu()
    from tests.gpu_helpers import hip_forward
    orc.set_threads(os.cpu_count() or 1)
    cloud = syn.scene_1k(P=600_000, seed=33)
    cloud._scaling -= 1.2
    cam = syn.camera_1k(4096, 2160, 70.0)
    for variant in ("pcheck_obb", "fov_pcheck_obb"):
        fov = syn.foveation_layers(cloud, seed=34) if variant == "fov_pcheck_obb" else None
        scene, cd = scene_dict(cloud, variant, fov), cam_dict(cam, gaze=(0.45, 0.5))
        want = orc.forward(variant, scene, cd)
        got = hip_forward(variant, scene, cd, debug=False)
        assert got["num_rendered"] == want["num_rendered"] and want["num_rendered"] > 500_000
        np.testing.assert_array_equal(got["radii"], want["radii"])
        np.testing.assert_array_equal(got["ranges"], want["ranges"])
        np.testing.assert_array_equal(got["point_list"], want["point_list"])
        check_image(got["color"], want["color"], name=variant + " 4K, 600 k Gaussians")
    orc.set_threads(1)


def test_8k_tile_grid_beyond_the_packed_tile_scan():
    """7680x4320 -> 129 600 tiles: above the 65 535 tiles the tile scan's packed 16-bit positions hold (it takes its
    LDS-atomics form there; tile_scan.h), and above every LDS table of the binning kernels."""
    _need_gpu()
    from tests.gpu_helpers import hip_forward
    cloud = syn.scene_1k(P=1500, seed=12)
    cloud._scaling += 0.5
    cam = syn.camera_1k(7680, 4320, 70.0)
    scene, cd = scene_dict(cloud, "pcheck_obb", None), cam_dict(cam, gaze=(0.6, 0.4))
    want = orc.forward("pcheck_obb", scene, cd)
    got = hip_forward("pcheck_obb", scene, cd)
    assert got["num_rendered"] == want["num_rendered"]
    np.testing.assert_array_equal(got["ranges"], want["ranges"])
    np.testing.assert_array_equal(got["point_list"], want["point_list"])
    check_image(got["color"], want["color"])


def test_autograd_module_end_to_end():
    """render() entry point + autograd: grads reach the raw parameters and match the oracle chain."""
    _need_gpu()
    from fov3dgs_amd.gaussian_renderer import render
    dev = "cuda:0"
    cloud = syn.scene_1k(P=500, seed=4).to(dev).requires_grad_(True)
    cam = syn.camera_1k(128, 96).to(dev)

    class Pipe:
        debug = True
    bg = torch.tensor([0.2, 0.3, 0.1], device=dev)
    out = render(cam, cloud, Pipe(), bg, cuda_type="pcheck_obb_sum")
    img = out["render"]
    assert img.shape == (3, 96, 128) and set(out) >= {"viewspace_points", "visibility_filter", "radii", "gs_count", "contribs"}
    w = torch.randn_like(img)
    (img * w).sum().backward()
    assert out["viewspace_points"].grad is not None and out["viewspace_points"].grad.abs().sum() > 0
    # oracle: same activations on CPU, chain rule through them with torch
    cpu = syn.scene_1k(P=500, seed=4).requires_grad_(True)
    scene = scene_dict(cpu, "pcheck_obb_sum")
    cd = cam_dict(cam.to("cpu"), bg=(0.2, 0.3, 0.1))
    cam.to(dev)
    of = orc.forward("pcheck_obb_sum", scene, cd)
    check_image(img.detach().cpu().numpy(), of["color"])
    og = orc.backward("pcheck_obb_sum", scene, cd, of, w.cpu().numpy())
    check_grad(cloud._xyz.grad.cpu().numpy(), og["dL_dmean3D"], "xyz")
    # opacity: d/d(logit) = dL_dopacity * sigmoid'
    s = torch.sigmoid(cpu._opacity.detach())
    check_grad(cloud._opacity.grad.cpu().numpy(), og["dL_dopacity"] * (s * (1 - s)).numpy(), "opacity")
    check_grad(cloud._scaling.grad.cpu().numpy(), og["dL_dscale"] * torch.exp(cpu._scaling.detach()).numpy(), "scaling")
    check_grad(cloud._features_dc.grad.cpu().numpy(), og["dL_dsh"][:, :1], "f_dc")
    check_grad(cloud._features_rest.grad.cpu().numpy(), og["dL_dsh"][:, 1:], "f_rest")


@pytest.mark.parametrize("sh_degree", (3, 1))
def test_row_sparse_gradients_equal_the_dense_ones(sh_degree):
    """fr_backward_args.row_sparse (extension; the reference returns dense tensors it zero-fills, rasterize_points.cu:171-179):
    the backward pass writes compact rows -- one per cull survivor, nothing cleared -- and render() hands autograd sparse tensors.
    Densified they equal the dense call's gradients (same kernels, same rows; two backward passes differ by the order their float
    atomics retire in), every Gaussian with a non-zero dense gradient row is among the sparse rows, and rows outside are exact zeros."""
    _need_gpu()
    from fov3dgs_amd.gaussian_renderer import render
    dev = "cuda:0"
    cam = syn.camera_1k(200, 136).to(dev)
    bg = torch.tensor([0.1, 0.0, 0.2], device=dev)

    class Pipe:
        debug = False
    w = None
    res = []
    for sparse in (False, True):
        cloud = small_cloud(P=4000, seed=29).to(dev).requires_grad_(True)
        cloud.fuse_activations = True   # raw parameters: every rasterizer input is a leaf
        cloud.row_sparse_grads = sparse
        cloud.active_sh_degree = sh_degree
        out = render(cam, cloud, Pipe(), bg, cuda_type="pcheck_obb_sum")
        if w is None:
            w = torch.randn_like(out["render"])
        (out["render"] * w).sum().backward()
        grads = dict(xyz=cloud._xyz.grad, scaling=cloud._scaling.grad, rotation=cloud._rotation.grad, opacity=cloud._opacity.grad,
                     f_dc=cloud._features_dc.grad, f_rest=cloud._features_rest.grad, screen=out["viewspace_points"].grad)
        assert all(g is not None for g in grads.values())
        if sparse:
            assert all(g.is_sparse for g in grads.values())
            rows = grads["xyz"].coalesce().indices()[0]
            assert 500 < rows.numel() < 4000 and bool((rows[1:] > rows[:-1]).all())
            res.append(({k: g.to_dense().cpu().numpy() for k, g in grads.items()}, rows.cpu().numpy(), out["radii"].cpu().numpy()))
        else:
            assert not any(g.is_sparse for g in grads.values())
            res.append(({k: g.cpu().numpy() for k, g in grads.items()}, None, out["radii"].cpu().numpy()))
    (dense, _, radii), (sp, rows, radii2) = res
    np.testing.assert_array_equal(radii, radii2)
    in_rows = np.zeros(4000, bool)
    in_rows[rows] = True
    assert in_rows[radii > 0].all()
    for k in dense:
        check_grad(sp[k], dense[k], "row-sparse " + k)
        assert np.abs(dense[k]).max() > 0, k
        assert not np.abs(dense[k].reshape(4000, -1))[~in_rows].any() and not np.abs(sp[k].reshape(4000, -1))[~in_rows].any()
        if sh_degree < 3 and k == "f_rest":
            assert not sp[k][:, (sh_degree + 1) ** 2 - 1:].any()  # coefficients beyond the active degree: zeros, written explicitly


@pytest.mark.parametrize("P", (20_037, 64, 33))
def test_narrow_gradient_tensors_with_clustered_visibility(P):
    """The narrow dense gradient tensors (positions, opacity, scales, rotations, DC coefficients of split SH storage) are written in whole
    lines: k_preprocess_bwd stores a wave's visible rows TOGETHER WITH the zeros between them, k_fill_groups clears the 32-row groups
    without a visible Gaussian (csrc/backward.hip SmallSet). A model whose visible Gaussians come in index runs -- thousands of rows
    behind the camera, runs of exactly one group, single visible rows far apart (a wave's 64 rows then span more than the 512 it puts
    together at a time), a last group cut off by P -- against the row-sparse call of the same step, which stores every row on its own.
    The tensors start as NaN (tests/conftest.py): every element has to be written by exactly the right kernel."""
    _need_gpu()
    from fov3dgs_amd.gaussian_renderer import render
    dev = "cuda:0"
    cam = syn.camera_1k(200, 136).to(dev)
    bg = torch.tensor([0.1, 0.0, 0.2], device=dev)

    class Pipe:
        debug = False
    hidden = np.zeros(P, bool)
    if P > 1000:
        hidden[:3000] = True            # the head of every tensor: empty groups only
        hidden[5000:5032] = True        # exactly one group (5000 = 32 * 156.25: it straddles two)
        hidden[5056:5088] = True        # exactly one aligned group
        hidden[9000:17000] = True       # a long run ...
        hidden[9000:17000:701] = False  # ... with single visible rows 701 apart
        hidden[P - 50:P - 3] = True     # the cut-off last group holds three visible rows at its end
    else:
        hidden[P // 2:] = True          # P = 64: one empty group behind a visible one; P = 33: the one-row last group is empty
    w = None
    res = []
    for sparse in (False, True):
        cloud = small_cloud(P=max(P, 8), seed=31)
        with torch.no_grad():
            cloud._xyz[torch.from_numpy(hidden)[: cloud._xyz.shape[0]], 2] = -6.0  # behind the camera
        cloud = cloud.to(dev).requires_grad_(True)
        cloud.fuse_activations = True
        cloud.row_sparse_grads = sparse
        out = render(cam, cloud, Pipe(), bg, cuda_type="pcheck_obb_sum")
        if w is None:
            w = torch.randn_like(out["render"])
        (out["render"] * w).sum().backward()
        grads = dict(xyz=cloud._xyz.grad, scaling=cloud._scaling.grad, rotation=cloud._rotation.grad, opacity=cloud._opacity.grad,
                     f_dc=cloud._features_dc.grad, f_rest=cloud._features_rest.grad, screen=out["viewspace_points"].grad)
        res.append(({k: (g.to_dense() if g.is_sparse else g).cpu().numpy() for k, g in grads.items()}, out["radii"].cpu().numpy()))
    (dense, radii), (sp, radii2) = res
    np.testing.assert_array_equal(radii, radii2)
    assert not (radii[hidden] > 0).any() and (radii[~hidden] > 0).sum() > (~hidden).sum() // 8
    for k in dense:
        assert np.isfinite(dense[k]).all(), k
        check_grad(dense[k], sp[k], "clustered visibility, dense vs row-sparse " + k)
        assert not np.abs(dense[k].reshape(P, -1))[radii == 0].any(), k
        assert np.abs(dense[k]).max() > 0, k


@pytest.mark.parametrize("P", (20000, 777))
def test_backward_in_ranges_of_rows(P):
    """fr_backward_args.num_ranges / range_done (round 6; rasterizer.GRADIENT_RANGE_HOOK): the per-Gaussian half of the backward call in
    four pieces over increasing ranges of rows, the host told behind each piece. The ranges tile [0, P) in order, start on multiples of
    32 (a group of the whole-line scheme never straddles two) and the gradients -- clustered visibility, every tensor NaN before the
    call -- are those of the call in one piece. A tensor snapshot taken INSIDE the callback (a copy enqueued on the stream right
    there) already holds the final rows of its range: that is what lets a communication stream start on them."""
    _need_gpu()
    from fov3dgs_amd import rasterizer as rz
    from fov3dgs_amd.gaussian_renderer import render
    dev = "cuda:0"
    cam = syn.camera_1k(200, 136).to(dev)
    bg = torch.tensor([0.1, 0.0, 0.2], device=dev)

    class Pipe:
        debug = False
    hidden = np.zeros(P, bool)
    if P > 1000:
        hidden[:3000] = True
        hidden[4990:5100] = True        # a hidden run across the first range's end (P / 4 = 5000 -> 4992)
        hidden[9000:17000] = True       # two whole ranges with ...
        hidden[9000:17000:701] = False  # ... single visible rows 701 apart
    w = None
    res = []
    for ranged in (False, True):
        cloud = small_cloud(P=P, seed=31)
        with torch.no_grad():
            cloud._xyz[torch.from_numpy(hidden), 2] = -6.0
        cloud = cloud.to(dev).requires_grad_(True)
        cloud.fuse_activations = True
        out = render(cam, cloud, Pipe(), bg, cuda_type="pcheck_obb_sum")
        if w is None:
            w = torch.randn_like(out["render"])
        calls, snaps = [], []

        def hook(k, lo, hi, grads):
            calls.append((k, lo, hi))
            snaps.append({n: t[lo:hi].clone() for n, t in grads.items() if t is not None})
        rz.GRADIENT_RANGE_HOOK, rz.GRADIENT_RANGES = (hook if ranged else None), 4
        try:
            (out["render"] * w).sum().backward()
        finally:
            rz.GRADIENT_RANGE_HOOK = None
        torch.cuda.synchronize()
        grads = dict(means3D=cloud._xyz.grad, scales=cloud._scaling.grad, rotations=cloud._rotation.grad, opacities=cloud._opacity.grad,
                     sh=cloud._features_dc.grad, sh_rest=cloud._features_rest.grad)
        res.append(({k: g.clone() for k, g in grads.items()}, calls, snaps))
    (one, c0, _), (four, calls, snaps) = res
    assert c0 == [] and [c[0] for c in calls] == [0, 1, 2, 3]
    assert calls[0][1] == 0 and calls[-1][2] == P and all(a[2] == b[1] for a, b in zip(calls, calls[1:]))
    assert all(lo % 32 == 0 for _, lo, _ in calls) and all(hi > lo for _, lo, hi in calls)
    for k in one:
        assert torch.isfinite(four[k]).all(), k
        check_grad(four[k].cpu().numpy(), one[k].cpu().numpy(), "backward in four ranges vs one piece: " + k)
        for (kk, lo, hi), snap in zip(calls, snaps):
            assert torch.equal(snap[k], four[k][lo:hi]), (k, kk)   # the rows of a range were final when the host was told


@pytest.mark.parametrize("variant", ["original", "pcheck_obb_sum"])
def test_raw_parameters_in_the_kernels_match_activate_then_render(variant):
    """fr_forward_args.raw_activations: exp / normalize / sigmoid applied inside the kernels give the image of
    activating first bit for bit (same device expressions), and the backward pass returns the gradients w.r.t. the raw
    parameters that the activation pass (itself checked against torch in tests/test_loss.py) would return."""
    _need_gpu()
    from fov3dgs_amd.activations import activate
    from fov3dgs_amd.gaussian_wrapper import get_gs_rasterizer
    from fov3dgs_amd.rasterizer import GaussianRasterizationSettings
    dev = "cuda:0"
    cam = syn.camera_1k(200, 136).to(dev)
    rs = GaussianRasterizationSettings(136, 200, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
                                       torch.tensor([0.1, 0.0, 0.2], device=dev), 1.0, cam.world_view_transform,
                                       cam.full_proj_transform, 3, cam.camera_center, False, False)
    cuda_type = {"original": "original", "pcheck_obb_sum": "pcheck_obb_sum"}[variant]
    w = None
    res = []
    for raw in (False, True):
        cloud = syn.scene_1k(P=2500, seed=11).to(dev)
        with torch.no_grad():
            cloud._rotation.mul_(torch.linspace(0.3, 4.0, 2500, device=dev).unsqueeze(1))  # far from unit length
        cloud.requires_grad_(True)
        if raw:
            s, q, o = cloud._scaling, cloud._rotation, cloud._opacity
        else:
            s, q, o = activate(cloud._scaling, cloud._rotation, cloud._opacity)
        out = get_gs_rasterizer(cuda_type, rs)(means3D=cloud.get_xyz, means2D=torch.zeros_like(cloud.get_xyz, requires_grad=True),
                                               opacities=o, shs=cloud.get_features_split, scales=s, rotations=q,
                                               **({"raw_activations": True} if raw else {}))
        img = out[0]
        if w is None:
            w = torch.randn_like(img)
        (img * w).sum().backward()
        res.append(dict(img=img.detach().cpu().numpy(), radii=out[1].cpu().numpy(), xyz=cloud._xyz.grad.cpu().numpy(),
                        scaling=cloud._scaling.grad.cpu().numpy(), rotation=cloud._rotation.grad.cpu().numpy(),
                        opacity=cloud._opacity.grad.cpu().numpy(), f_dc=cloud._features_dc.grad.cpu().numpy()))
    np.testing.assert_array_equal(res[0]["img"], res[1]["img"])
    np.testing.assert_array_equal(res[0]["radii"], res[1]["radii"])
    assert (res[0]["radii"] > 0).sum() > 500
    for name in ("xyz", "scaling", "rotation", "opacity", "f_dc"):
        assert np.abs(res[0][name]).max() > 0
        check_grad(res[1][name], res[0][name], "raw " + name)
    # the rasterizer refuses the combination it has no kernels for
    with pytest.raises(Exception):
        get_gs_rasterizer(cuda_type, rs)(means3D=cloud.get_xyz, means2D=torch.zeros_like(cloud.get_xyz), opacities=cloud._opacity,
                                         shs=cloud.get_features, cov3D_precomp=torch.zeros(2500, 6, device=dev), raw_activations=True)


def test_split_sh_storage_matches_concatenated():
    """fr_forward_args.shs_rest: features_dc / features_rest handed over as stored give the same image (bit for bit:
    same summation order) and the same gradients as the torch.cat'ed [P,16,3] tensor of the reference interface."""
    _need_gpu()
    from fov3dgs_amd.diff_gaussian_rasterization_pcheck_obb_sum import GaussianRasterizationSettings, GaussianRasterizer
    dev = "cuda:0"
    cam = syn.camera_1k(160, 112).to(dev)
    rs = GaussianRasterizationSettings(112, 160, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
                                       torch.tensor([0.1, 0.0, 0.2], device=dev), 1.0, cam.world_view_transform,
                                       cam.full_proj_transform, 3, cam.camera_center, False, False)
    w = None
    res = []
    for split in (False, True):
        cloud = syn.scene_1k(P=900, seed=9).to(dev).requires_grad_(True)
        shs = cloud.get_features_split if split else cloud.get_features
        out = GaussianRasterizer(rs)(means3D=cloud.get_xyz, means2D=torch.zeros_like(cloud.get_xyz, requires_grad=True),
                                     opacities=cloud.get_opacity, shs=shs, scales=cloud.get_scaling,
                                     rotations=cloud.get_rotation)
        img = out[0]
        if w is None:
            w = torch.randn_like(img)
        (img * w).sum().backward()
        res.append((img.detach().cpu().numpy(), cloud._features_dc.grad.cpu().numpy(), cloud._features_rest.grad.cpu().numpy(),
                    cloud._xyz.grad.cpu().numpy()))
    np.testing.assert_array_equal(res[0][0], res[1][0])
    for a, b, name in zip(res[0][1:], res[1][1:], ("f_dc", "f_rest", "xyz")):
        check_grad(b, a, name)


def test_packed_model_gives_identical_images():
    """fr_forward_args.packed_geom / packed_colour (the static-model layout of include/fovraster.h): the pack kernels
    lay the rows out as documented, and every variant renders bit-identical images, radii and statistics with them."""
    _need_gpu()
    from fov3dgs_amd.rasterizer import pack_model
    from fov3dgs_amd.gaussian_renderer import render
    from fov3dgs_amd.gaussian_renderer_fov import render as render_fov
    dev = "cuda:0"
    cloud = syn.scene_1k(P=1500, seed=12).to(dev)
    cam = syn.camera_1k(208, 144).to(dev)
    bg = torch.tensor([0.2, 0.1, 0.0], device=dev)

    class Pipe:
        debug = False
    with torch.no_grad():
        # plain variants: concatenated and split SH inputs pack to the same rows
        # render() takes the activations from cloud.get_activated (the fused pass): the packed copy must hold the same values
        act_s, act_q, act_o = cloud.get_activated
        pk = pack_model(cloud.get_xyz, act_s, act_q, act_o, shs=cloud.get_features)
        dc, rest = cloud.get_features_split
        pk2 = pack_model(cloud.get_xyz, act_s, act_q, act_o, shs=dc, shs_rest=rest)
        assert torch.equal(pk.geom, pk2.geom) and torch.equal(pk.colour, pk2.colour)
        g = pk.geom.cpu().numpy()
        np.testing.assert_array_equal(g[:, 0:3], cloud.get_xyz.cpu().numpy())
        np.testing.assert_array_equal(g[:, 3:6], act_s.cpu().numpy())
        np.testing.assert_array_equal(g[:, 6:10], act_q.cpu().numpy())
        np.testing.assert_array_equal(g[:, 12], act_o.cpu().numpy()[:, 0])
        assert not g[:, 10:12].any() and not g[:, 13:].any()
        cu = pk.cull.cpu().numpy()
        np.testing.assert_array_equal(cu[:, :3], g[:, 0:3])
        np.testing.assert_allclose(cu[:, 3], (g[:, 3:6].max(axis=1)) ** 2, rtol=1e-5)  # unit quaternions: factor 1
        c = pk.colour.cpu().numpy()
        f = cloud.get_features.cpu().numpy().reshape(-1, 48)
        np.testing.assert_array_equal(c[:, :45], f[:, 3:])
        np.testing.assert_array_equal(c[:, 45:48], f[:, :3])
        assert not c[:, 48:].any()
        for cuda_type in ("original", "pcheck_obb", "pcheck_obb_sum"):
            a = render(cam, cloud, Pipe(), bg, cuda_type=cuda_type)
            b = render(cam, cloud, Pipe(), bg, cuda_type=cuda_type, packed=pk)
            for k in a:
                if k == "contribs":  # float atomics: the summation order differs from run to run
                    np.testing.assert_allclose(a[k].cpu().numpy(), b[k].cpu().numpy(), rtol=1e-4, atol=1e-6)
                else:
                    assert torch.equal(a[k], b[k]), (cuda_type, k)
        # foveated
        highest, shs_dcs, opac = syn.foveation_layers(cloud, seed=3)
        pf = pack_model(cloud.get_xyz, act_s, act_q, opac, shs=cloud.get_rest_features,
                        shs_dcs=shs_dcs, highest_levels=highest)
        gf = pf.geom.cpu().numpy()
        np.testing.assert_array_equal(gf[:, 10], highest.cpu().numpy().reshape(-1))
        np.testing.assert_array_equal(gf[:, 12:16], opac.cpu().numpy())
        cf = pf.colour.cpu().numpy()
        np.testing.assert_array_equal(cf[:, :45], cloud.get_rest_features.cpu().numpy().reshape(-1, 45))
        np.testing.assert_array_equal(cf[:, 48:60], shs_dcs.cpu().numpy().reshape(-1, 12))
        for gaze in ((0.5, 0.5), (0.2, 0.8)):
            kw = dict(alpha=0.05, gazeArray=torch.tensor(gaze), blending=True, highest_levels=highest, shs_dcs=shs_dcs,
                      opacities=opac)
            a = render_fov(cam, cloud, bg, **kw)
            b = render_fov(cam, cloud, bg, packed=pf, **kw)
            assert torch.equal(a["render"], b["render"]) and torch.equal(a["radii"], b["radii"])
        # render() packs a static model by itself: the second call with the same, unmodified tensor objects
        class Frozen:
            pass
        fz = Frozen()
        fz.get_xyz, fz.get_scaling, fz.get_rotation = cloud.get_xyz, cloud.get_scaling, cloud.get_rotation
        fz.get_rest_features, fz.active_sh_degree = cloud.get_rest_features, cloud.active_sh_degree
        kw = dict(alpha=0.05, gazeArray=torch.tensor((0.4, 0.6)), blending=True, highest_levels=highest, shs_dcs=shs_dcs,
                  opacities=opac)
        from fov3dgs_amd.gaussian_renderer_fov import invalidate_packed
        ref_img = render_fov(cam, fz, bg, **kw)["render"]  # default: the reference's interface, nothing is cached
        assert not hasattr(fz, "_fovraster_pack_state")
        imgs = [render_fov(cam, fz, bg, packed="auto", **kw)["render"] for _ in range(3)]
        assert fz._fovraster_pack_state.packed is not None
        for im in imgs:
            assert torch.equal(im, ref_img)
        fz.get_scaling.mul_(1.5)  # an in-place change is noticed (version counter): no stale packed copy
        ref2 = render_fov(cam, fz, bg, **kw)["render"]
        assert not torch.equal(ref2, ref_img)
        assert torch.equal(render_fov(cam, fz, bg, packed="auto", **kw)["render"], ref2) and fz._fovraster_pack_state.packed is None
        assert torch.equal(render_fov(cam, fz, bg, packed="auto", **kw)["render"], ref2) and fz._fovraster_pack_state.packed is not None
        # `t.data = new tensor` keeps object and version but not the storage address: noticed too
        fz.get_scaling.data = fz.get_scaling.data * 0.5
        ref3 = render_fov(cam, fz, bg, **kw)["render"]
        assert torch.equal(render_fov(cam, fz, bg, packed="auto", **kw)["render"], ref3) and fz._fovraster_pack_state.packed is None
        assert torch.equal(render_fov(cam, fz, bg, packed="auto", **kw)["render"], ref3) and fz._fovraster_pack_state.packed is not None
        # a write through .data is invisible to the version counter: the documented remedy is invalidate_packed()
        fz.get_scaling.data.mul_(1.25)
        invalidate_packed(fz)
        ref4 = render_fov(cam, fz, bg, **kw)["render"]
        assert not torch.equal(ref4, ref3)
        assert torch.equal(render_fov(cam, fz, bg, packed="auto", **kw)["render"], ref4)
        # a model whose getters build new tensors per call is never packed
        for _ in range(3):
            render_fov(cam, cloud, bg, packed="auto", **kw)
        assert cloud._fovraster_pack_state.packed is None
        # a packed model of the wrong size is refused
        with pytest.raises(RuntimeError):
            render(cam, syn.scene_1k(P=10, seed=1).to(dev), Pipe(), bg, cuda_type="pcheck_obb", packed=pk)


def test_foveated_render_entry_point():
    _need_gpu()
    from fov3dgs_amd.gaussian_renderer_fov import render
    dev = "cuda:0"
    cloud = syn.scene_1k(P=800, seed=6).to(dev)
    cam = syn.camera_1k(160, 128).to(dev)
    highest, shs_dcs, opac = syn.foveation_layers(cloud, seed=8)
    bg = torch.zeros(3, device=dev)
    with torch.no_grad():
        out = render(cam, cloud, bg, alpha=0.05, gazeArray=torch.tensor([0.3, 0.6]), blending=True,
                     highest_levels=highest, shs_dcs=shs_dcs, opacities=opac)
    cpu = cloud.to("cpu")
    scene = scene_dict(cpu, "fov_pcheck_obb", (highest.cpu(), shs_dcs.cpu(), opac.cpu()))
    cd = cam_dict(cam.to("cpu"), gaze=(0.3, 0.6), alpha=0.05)
    want = orc.forward("fov_pcheck_obb", scene, cd)
    check_image(out["render"].cpu().numpy(), want["color"])
    np.testing.assert_array_equal(out["radii"].cpu().numpy(), want["radii"])


@pytest.mark.parametrize("variant", ("fov_pcheck_obb", "pcheck_obb"))
def test_full_size_properties(variant):
    """BASELINE-sized frame (6 M Gaussians, 1080p), too large for the oracle: size-independent properties.
    The tile ranges partition [0, num_rendered); every tile list is sorted by (depth bits, Gaussian index) -- the
    reference's stable radix order; listed Gaussians are visible; two runs are bit-identical; the plain inference
    variant renders exactly what the training variant (same culling, extra statistics) renders."""
    _need_gpu()
    from fov3dgs_amd import _native, rasterizer as rz
    lib = _native.load()
    dev = torch.device("cuda", 0)
    cloud = syn.scene_bicycle_scale(P=6_000_000, seed=1)
    fov = [t.to(dev) for t in syn.foveation_layers(cloud, seed=2)] if variant == "fov_pcheck_obb" else None
    cloud = cloud.to(dev)
    cam = syn.camera_ring(1, 8).to(dev)
    W, H = cam.image_width, cam.image_height
    T = ((W + 15) // 16) * ((H + 15) // 16)
    rs = rz.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=dev),
                                          1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
    E = torch.Tensor([])
    vid = _native.VARIANT_IDS[variant]
    with torch.no_grad():
        xyz, sc, rot = cloud.get_xyz, cloud.get_scaling.contiguous(), cloud.get_rotation.contiguous()

        def run(v):
            if v == _native.VARIANT_IDS["fov_pcheck_obb"]:
                return rz._forward_native(v, rs, xyz, cloud.get_rest_features.contiguous(), E, fov[2], sc, rot, E, fov[1], fov[0],
                                          (0.31, 0.62), 0.05)
            return rz._forward_native(v, rs, xyz, cloud._features_dc, E, cloud.get_opacity, sc, rot, E,
                                      sh_rest=cloud._features_rest)
        r = run(vid)
        torch.cuda.synchronize()
    D, color, radii, geom, binb, img = r[:6]
    assert D > 1_000_000 and torch.isfinite(color).all()

    def view(buf, ptr, count, dtype):
        off = ptr - buf.data_ptr()
        return buf[off:off + 4 * count].view(dtype)
    ranges = view(img, lib.fr_image_ranges(vid, W, H, img.data_ptr()), 2 * T, torch.int32).view(T, 2).long()
    from tests.gpu_helpers import vis_list_of
    vis = vis_list_of(lib, vid, xyz.shape[0], geom)
    assert bool((vis[1:] > vis[:-1]).all()), "the list of cull survivors is in index order"
    items = view(binb, lib.fr_binning_point_list(vid, D, binb.data_ptr()), D, torch.int32).long()  # positions in that list
    plist = vis[items]
    rec = view(geom, lib.fr_geometry_records(vid, xyz.shape[0], geom.data_ptr()), 12 * xyz.shape[0], torch.float32).view(-1, 12)  # per item
    # ranges: non-empty tiles tile [0, D) exactly once
    n = ranges[:, 1] - ranges[:, 0]
    assert int(n.sum()) == D and int(n.min()) >= 0
    ne = ranges[n > 0]
    order = torch.argsort(ne[:, 0])
    ne = ne[order]
    assert int(ne[0, 0]) == 0 and int(ne[-1, 1]) == D and bool((ne[1:, 0] == ne[:-1, 1]).all())
    # listed Gaussians are visible, every visible Gaussian is listed
    assert bool((radii[plist] > 0).all())
    seen = torch.zeros(xyz.shape[0], dtype=torch.bool, device=dev)
    seen[plist] = True
    assert bool((seen == (radii > 0)).all())
    # per-tile order: key = (depth bits, index) strictly increasing inside every tile
    depth_bits = rec[items, 9].contiguous().view(torch.int32).long()
    assert bool((rec[items, 11].contiguous().view(torch.int32).long() == plist).all()), "a record carries its Gaussian's index"
    key = depth_bits * (1 << 32) + plist
    tile_of = torch.repeat_interleave(torch.arange(T, device=dev), n)  # entries are laid out tile by tile in range order?
    start = torch.zeros(D, dtype=torch.bool, device=dev)
    start[ranges[n > 0, 0]] = True
    inc = key[1:] > key[:-1]
    assert bool((inc | start[1:]).all()), "a tile list is not sorted by (depth, index)"
    del tile_of
    # determinism
    img1 = color.clone(); pl1 = plist.clone()
    with torch.no_grad():
        r2 = run(vid)
        torch.cuda.synchronize()
    assert r2[0] == D and torch.equal(r2[1], img1)
    plist2 = vis_list_of(lib, vid, xyz.shape[0], r2[3])[view(r2[4], lib.fr_binning_point_list(vid, D, r2[4].data_ptr()), D, torch.int32).long()]
    assert torch.equal(plist2, pl1)
    if variant == "pcheck_obb":
        with torch.no_grad():
            r3 = run(_native.VARIANT_IDS["pcheck_obb_sum"])
            torch.cuda.synchronize()
        assert r3[0] == D and torch.equal(r3[1], img1)


def test_randomised_sweep():
    """Seeded random sweep over sizes, image shapes, splat size mixes, gaze positions and variants (edge sizes 1, 2,
    63..65 included): instance lists bit-exact, images within the tolerance of check_image."""
    _need_gpu()
    from tests.gpu_helpers import hip_forward
    rng = np.random.default_rng(7)
    variants = ("original", "pcheck_obb_sum", "pcheck_obb", "fov_pcheck_obb", "pcheck_obb_max")
    for r in range(15):
        variant = variants[r % len(variants)]
        P = int(rng.choice([1, 2, 63, 64, 65, 500, 3000, 9000]))
        W, H = int(rng.integers(17, 700)), int(rng.integers(17, 500))
        cloud = small_cloud(P, seed=int(rng.integers(1 << 30)), big_fraction=float(rng.choice([0.0, 0.1, 0.5]))) if P >= 8 \
            else syn.scene_1k(P=P, seed=r)
        if rng.random() < 0.4 and P >= 8:
            cloud._scaling[: max(1, P // 50)] += 3.0  # a few frame-filling splats
        cam = small_camera(W, H)
        fov = syn.foveation_layers(cloud, seed=r) if variant == "fov_pcheck_obb" else None
        scene = scene_dict(cloud, variant, fov)
        cd = cam_dict(cam, gaze=(float(rng.uniform(-0.3, 1.3)), float(rng.uniform(-0.3, 1.3))), alpha=float(rng.choice([0.05, 0.02, 0.2])))
        want = orc.forward(variant, scene, cd)
        got = hip_forward(variant, scene, cd)
        tag = f"round {r}: {variant} P={P} {W}x{H}"
        assert got["num_rendered"] == want["num_rendered"], tag
        np.testing.assert_array_equal(got["radii"], want["radii"], err_msg=tag)
        np.testing.assert_array_equal(got["ranges"], want["ranges"], err_msg=tag)
        np.testing.assert_array_equal(got["point_list"], want["point_list"], err_msg=tag)
        check_image(got["color"], want["color"], name=tag)
        if P >= 1 and scene.get("scales") is not None and scene.get("shs") is not None:
            # the packed static-model layout: bit-identical to the ordinary tensors
            pk = hip_forward(variant, scene, cd, packed=True)
            assert pk["num_rendered"] == got["num_rendered"], tag
            for k in ("radii", "ranges", "point_list", "color"):
                np.testing.assert_array_equal(pk[k], got[k], err_msg=tag + " packed " + k)


def test_mark_visible_matches_oracle():
    """fr_mark_visible / GaussianRasterizer.markVisible (R0 rasterizer_impl.cu:54-66, rasterize_points.cu:198-217)
    against the oracle, including points exactly on the near threshold p_view.z == 0.2 (culled: the test is <=)."""
    _need_gpu()
    from fov3dgs_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = "cuda:0"
    cloud = small_cloud(5000, seed=31)
    cam = syn.camera_1k(128, 128)  # identity world-to-camera: p_view.z == z bit for bit
    xyz = cloud.get_xyz.detach().numpy().copy()
    z02 = np.float32(0.2)
    xyz[:300, 2] = z02                                    # exactly on the threshold
    xyz[300:600, 2] = np.nextafter(z02, np.float32(1))    # one ulp in front of it
    xyz[600:900, 2] = np.nextafter(z02, np.float32(0))    # one ulp behind it
    xyz[900:1000, 2] = np.float32("nan")
    cd = cam_dict(cam)
    want = orc.mark_visible(dict(means3D=xyz, opacities=np.zeros((len(xyz), 1), np.float32)), cd)
    assert not want[:300].any() and want[300:600].all() and not want[600:900].any()
    cam.to(dev)
    rs = GaussianRasterizationSettings(128, 128, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=dev), 1.0,
                                       cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
    got = GaussianRasterizer(rs).markVisible(torch.as_tensor(xyz).to(dev))
    assert got.dtype == torch.bool and got.shape == (len(xyz),)
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    # a tilted camera (general view matrix) and the empty cloud
    cam2 = small_camera(160, 96)
    cd2 = cam_dict(cam2)
    want2 = orc.mark_visible(dict(means3D=xyz[1000:], opacities=np.zeros((len(xyz) - 1000, 1), np.float32)), cd2)
    cam2.to(dev)
    rs2 = rs._replace(viewmatrix=cam2.world_view_transform, projmatrix=cam2.full_proj_transform)
    got2 = GaussianRasterizer(rs2).markVisible(torch.as_tensor(xyz[1000:]).to(dev))
    np.testing.assert_array_equal(got2.cpu().numpy(), want2)
    assert 0 < want2.sum() < len(want2)
    assert GaussianRasterizer(rs).markVisible(torch.zeros((0, 3), device=dev)).shape == (0,)


@pytest.mark.parametrize("gaze", ((0.5, 0.5), (0.15, 0.8)))
def test_fov_level_colours_match_oracle(gaze):
    """compute_fov_colors (RF rasterizer_impl.cu:490-530) directly: the per-level (r, g, b, opacity) rows and the
    level ranges the binning kernel leaves for every visible Gaussian against the oracle's fov_colors /
    level_ranges, for exactly the slots in a Gaussian's level range (the others are unwritten in the reference too)."""
    _need_gpu()
    from tests.gpu_helpers import hip_forward
    from fov3dgs_amd import _native
    lib = _native.load()
    scene, cam = small_case("fov_pcheck_obb", P=6000, seed=17, gaze=gaze, width=640, height=400)
    want = orc.forward("fov_pcheck_obb", scene, cam)
    for packed in (False, True):
        got = hip_forward("fov_pcheck_obb", scene, cam, packed=packed)
        geom = got["_buffers"][0]
        P = scene["means3D"].shape[0]

        def view(ptr, count, dtype):
            off = ptr - geom.data_ptr()
            return geom[off:off + 4 * count].view(dtype)
        # both are kept per ITEM (position in the list of cull survivors): spread them out by Gaussian index
        vl = got["vis_list"].astype(np.int64)
        lvl = np.full((P, 4, 4), np.nan, np.float32)
        lvl[vl] = view(lib.fr_geometry_level_colours(P, geom.data_ptr()), 16 * P, torch.float32).view(P, 4, 4).cpu().numpy()[:len(vl)]
        lr = np.zeros(P, np.int32)
        lr[vl] = view(lib.fr_geometry_level_ranges(P, geom.data_ptr()), P, torch.int32).cpu().numpy()[:len(vl)]
        vis = want["radii"] > 0
        assert vis.sum() > 1000
        np.testing.assert_array_equal((lr & 0xff)[vis], want["level_ranges"][vis, 0])
        np.testing.assert_array_equal(((lr >> 8) & 0xff)[vis], want["level_ranges"][vis, 1])
        lo, hi = want["level_ranges"][:, 0], want["level_ranges"][:, 1]
        seen_levels = set()
        for l in range(4):
            m = vis & (lo <= l) & (l <= hi)
            if not m.any():
                continue
            seen_levels.add(l)
            np.testing.assert_array_equal(lvl[m, l, 3], scene["opacities"][m, l])
            np.testing.assert_allclose(lvl[m, l, :3], want["fov_colors"][m, l], rtol=0, atol=1e-6)
            from tests import parity_report
            parity_report.record("colour", f"fov level colours level {l} gaze={gaze} packed={packed}", n=int(m.sum()),
                                 max_abs=float(np.abs(lvl[m, l, :3] - want["fov_colors"][m, l]).max()),
                                 bitwise_equal=float(np.mean(lvl[m, l, :3].view(np.uint32) == want["fov_colors"][m, l].view(np.uint32))))
        assert len(seen_levels) == 4


@pytest.mark.parametrize("gaze", ((0.4, 0.55), (0.9, 0.1)))
def test_shared_model_foveated_baseline(gaze):
    """the SMFR baseline (…_naive_pcheck_obb) -- lists bit-exact, image within tolerance, and the
    gaussian_renderer_fov_naive.render() entry point."""
    _need_gpu()
    from tests.gpu_helpers import hip_forward
    scene_f, cam = small_case("fov_pcheck_obb", P=6000, seed=23, gaze=gaze, width=640, height=400)
    plain, _ = small_case("pcheck_obb", P=6000, seed=23, width=640, height=400)
    scene = dict(plain, highest_levels=scene_f["highest_levels"])
    want = orc.forward("naive_pcheck_obb", scene, cam)
    assert want["tile_blend"].sum() > 30
    got = hip_forward("naive_pcheck_obb", scene, cam)
    assert got["num_rendered"] == want["num_rendered"]
    np.testing.assert_array_equal(got["radii"], want["radii"])
    np.testing.assert_array_equal(got["ranges"], want["ranges"])
    np.testing.assert_array_equal(got["point_list"], want["point_list"])
    check_image(got["color"], want["color"], name=f"SMFR gaze={gaze}")
    # entry point
    from fov3dgs_amd.gaussian_renderer_fov_naive import render
    dev = "cuda:0"
    cloud = small_cloud(6000, 23).to(dev)
    camo = small_camera(640, 400).to(dev)
    with torch.no_grad():
        out = render(camo, cloud, torch.tensor([0.1, 0.2, 0.3], device=dev), alpha=0.05, gazeArray=torch.tensor(gaze), blending=True,
                     highest_levels=torch.as_tensor(scene["highest_levels"]).to(dev))
    assert set(out) == {"render", "viewspace_points", "visibility_filter", "radii"}
    # (the model's activations run on the GPU here: a few radii may differ in the last place of a scale, see S6M in
    # tests/test_full_size_parity.py; the image stays within tolerance)
    check_image(out["render"].cpu().numpy(), want["color"], name=f"SMFR render() gaze={gaze}")


def test_multi_model_foveated_baseline():
    """the MMFR baseline (…_mmfr_pcheck_obb) -- every level's render against the oracle (lists
    bit-exact, image within tolerance), and gaussian_renderer_fov_mmfr.render() = the sum of the level renders."""
    _need_gpu()
    from tests.gpu_helpers import hip_forward
    gaze = (0.35, 0.6)
    levels = []
    for level in range(4):
        scene, cam = small_case("pcheck_obb", P=5000, seed=40 + level, gaze=gaze, width=640, height=400)  # a model per level
        scene = dict(scene, highest_levels=np.zeros((5000, 1), np.float32))
        cam = dict(cam, cur_level=float(level))
        want = orc.forward("mmfr_pcheck_obb", scene, cam)
        got = hip_forward("mmfr_pcheck_obb", scene, cam)
        tag = f"MMFR level {level}"
        assert got["num_rendered"] == want["num_rendered"] > 0, tag
        np.testing.assert_array_equal(got["radii"], want["radii"], err_msg=tag)
        np.testing.assert_array_equal(got["ranges"], want["ranges"], err_msg=tag)
        np.testing.assert_array_equal(got["point_list"], want["point_list"], err_msg=tag)
        check_image(got["color"], want["color"], name=tag)
        levels.append(want["color"])
    from fov3dgs_amd.gaussian_renderer_fov_mmfr import render
    dev = "cuda:0"
    models = [small_cloud(5000, 40 + level).to(dev) for level in range(4)]
    camo = small_camera(640, 400).to(dev)
    with torch.no_grad():
        out = render(camo, torch.tensor([0.1, 0.2, 0.3], device=dev), alpha=0.05, gazeArray=torch.tensor(gaze), blending=True,
                     multi_gs=models, layer_num=4)
    check_image(out["render"].cpu().numpy(), sum(levels), name="MMFR render()")


def test_backward_twice_gives_the_same_gradients():
    """fr_backward is idempotent like the reference's (fresh zero tensors per call, rasterize_points.cu:171-179): a second
    backward pass over the same forward state (retain_graph, torch.autograd.grad twice) returns the same gradients, not
    twice the first ones (round 2 accumulated into sums only the forward pass cleared)."""
    _need_gpu()
    from fov3dgs_amd.gaussian_renderer import render
    dev = "cuda:0"
    cloud = syn.scene_1k(P=3000, seed=9).to(dev).requires_grad_(True)
    cam = small_camera(320, 200).to(dev)

    class Pipe:
        debug = False
    out = render(cam, cloud, Pipe(), torch.tensor([0.2, 0.3, 0.1], device=dev), cuda_type="pcheck_obb_sum")
    w = torch.randn_like(out["render"])
    params = [cloud._xyz, cloud._scaling, cloud._rotation, cloud._opacity, cloud._features_dc, cloud._features_rest]
    loss = (out["render"] * w).sum()
    g1 = torch.autograd.grad(loss, params, retain_graph=True)
    g2 = torch.autograd.grad(loss, params, retain_graph=True)
    loss.backward()
    for p, a, b in zip(params, g1, g2):
        assert int((a.reshape(len(a), -1).abs().amax(dim=1) > 0).sum()) > 500
        # (the three passes add the same terms with float atomics in whatever order the hardware takes them)
        check_grad(b.cpu().numpy(), a.cpu().numpy(), "second autograd.grad vs first")
        check_grad(p.grad.cpu().numpy(), a.cpu().numpy(), "backward() after two autograd.grad calls")


def test_prefiltered_violation_is_reported():
    """`prefiltered` (GaussianRasterizationSettings field 11): the reference traps when a point of a cloud declared
    prefiltered is culled by the near plane (cuda_rasterizer/auxiliary.h:156-160); here fr_forward returns
    FR_ERR_PREFILTERED with the reference's message. A cloud that keeps the promise renders as without the flag."""
    _need_gpu()
    from fov3dgs_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = "cuda:0"
    cam = syn.camera_1k(128, 128).to(dev)
    cloud = syn.scene_1k(P=800, seed=2).to(dev)
    rs = GaussianRasterizationSettings(128, 128, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=dev), 1.0,
                                       cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, True, False)
    with torch.no_grad():
        xyz = cloud.get_xyz.clone()
        m2 = torch.zeros_like(xyz)
        kw = dict(shs=cloud.get_features, scales=cloud.get_scaling, rotations=cloud.get_rotation)
        assert (xyz[:, 2] > 0.2).all()
        img_pf, radii_pf = GaussianRasterizer(rs)(xyz, m2, cloud.get_opacity, **kw)
        img, radii = GaussianRasterizer(rs._replace(prefiltered=False))(xyz, m2, cloud.get_opacity, **kw)
        assert torch.equal(img, img_pf) and torch.equal(radii, radii_pf)
        xyz[17, 2] = 0.1  # behind the near plane
        with pytest.raises(RuntimeError, match="filtered although prefiltered is set"):
            GaussianRasterizer(rs)(xyz, m2, cloud.get_opacity, **kw)
        GaussianRasterizer(rs._replace(prefiltered=False))(xyz, m2, cloud.get_opacity, **kw)  # without the promise: culled, no error
    # the oracle restates the same rule
    cpu = syn.scene_1k(P=800, seed=2)
    scene = scene_dict(cpu, "original")
    scene["means3D"] = xyz.cpu().numpy()
    cd = dict(cam_dict(cam.to("cpu")), prefiltered=True)
    with pytest.raises(RuntimeError, match="prefiltered"):
        orc.forward("original", scene, cd)


@pytest.mark.parametrize("variant", ("pcheck_obb", "fov_pcheck_obb", "pcheck_obb_sum"))
def test_two_frames_in_flight(variant):
    """fr_forward_begin / fr_forward_finish (include/fovraster.h): the head of frame B is enqueued on a second stream before
    frame A's instance count has been waited for. Two different frames (two cameras' scale modifiers), interleaved
    begin(A) begin(B) finish(A) finish(B), three rounds over the same two workspace sets: images, lists and statistics equal the
    oracle's and the back-to-back calls' bit for bit."""
    _need_gpu()
    from tests.gpu_helpers import VARIANT_IDS, _t, settings_from, hip_forward
    from fov3dgs_amd.rasterizer import _forward_begin
    scene, cam_a = small_case(variant, P=5003, seed=19, width=408, height=232)
    cam_b = dict(cam_a, scale_modifier=1.7, gaze=(0.3, 0.6))
    wants = [orc.forward(variant, scene, c) for c in (cam_a, cam_b)]
    serial = [hip_forward(variant, scene, c, debug=False) for c in (cam_a, cam_b)]
    dev = "cuda:0"
    vid = VARIANT_IDS[variant]
    tens = {k: _t(scene.get(k), dev) for k in ("means3D", "shs", "colors_precomp", "opacities", "scales", "rotations",
                                                "cov3D_precomp", "shs_dcs", "highest_levels")}
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    torch.cuda.synchronize()

    def begin(cam, st):
        rs = settings_from(cam, dev, debug=False)
        with torch.cuda.stream(st):
            return _forward_begin(vid, rs, tens["means3D"], tens["shs"], tens["colors_precomp"], tens["opacities"], tens["scales"],
                                  tens["rotations"], tens["cov3D_precomp"], tens["shs_dcs"], tens["highest_levels"],
                                  cam.get("gaze", (0.5, 0.5)), cam.get("alpha", 0.05), persistent=True)
    for rnd in range(3):
        fa = begin(cam_a, streams[0])
        fb = begin(cam_b, streams[1])
        ra, rb = fa.finish(), fb.finish()
        torch.cuda.synchronize()
        for res, want, ser, name in ((ra, wants[0], serial[0], "A"), (rb, wants[1], serial[1], "B")):
            assert res[0] == want["num_rendered"]
            np.testing.assert_array_equal(res[2].cpu().numpy(), want["radii"])
            np.testing.assert_array_equal(res[1].cpu().numpy(), ser["color"])
            check_image(res[1].cpu().numpy(), want["color"], name=f"{variant} frame {name} of two in flight")
            if variant == "pcheck_obb_sum":
                np.testing.assert_array_equal(res[6].cpu().numpy(), want["gaussians_count"])
    # a frame abandoned between its halves releases its handle and leaves the streams usable
    fa = begin(cam_a, streams[0])
    del fa
    torch.cuda.synchronize()
    again = begin(cam_a, streams[0]).finish()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(again[1].cpu().numpy(), serial[0]["color"])


@pytest.mark.parametrize("variant", ("fov_pcheck_obb", "pcheck_obb_sum"))
def test_binning_workspace_requested_before_the_count(variant):
    """From the second frame of a kind (variant, P, W, H) on, fr_forward asks for the binning workspace BEFORE the instance
    count is in, sized like the largest frame of the kind so far plus a quarter (csrc/api.hip). Small frame (first call: sized
    exactly), small frame again (early request fits), a frame with > 1.6x the instances (early request too small: asked
    again), the small frame in the now oversized workspace -- lists and images as the oracle's each time, and the backward
    pass of the training variant finds its lists in a workspace laid out for more instances than the frame has."""
    _need_gpu()
    from tests.gpu_helpers import hip_forward, hip_backward
    scene, cam = small_case(variant, P=4801, seed=23, width=424, height=248)  # a (P, W, H) no other test uses
    big = dict(cam, scale_modifier=2.5)
    want, want_big = orc.forward(variant, scene, cam), orc.forward(variant, scene, big)
    assert want_big["num_rendered"] > 1.6 * want["num_rendered"]
    for cd, w in ((cam, want), (cam, want), (big, want_big), (cam, want)):
        got = hip_forward(variant, scene, cd, debug=False)
        assert got["num_rendered"] == w["num_rendered"]
        np.testing.assert_array_equal(got["ranges"], w["ranges"])
        np.testing.assert_array_equal(got["point_list"], w["point_list"])
        check_image(got["color"], w["color"], name=variant + " early workspace request")
    if variant == "pcheck_obb_sum":
        rng = np.random.default_rng(5)
        dL = rng.standard_normal(want["color"].shape).astype(np.float32)
        g = hip_backward(variant, got, dL)
        wb = orc.backward(variant, scene, cam, want, dL)
        for k in ("dL_dmean3D", "dL_dopacity", "dL_dscale", "dL_drot", "dL_dsh"):
            check_grad(g[k].reshape(wb[k].shape), wb[k], k + " (oversized binning workspace)")


@pytest.mark.parametrize("variant", ("fov_pcheck_obb", "naive_pcheck_obb"))
@pytest.mark.parametrize("levels", ("fractional", "out_of_range"))
def test_level_filter_with_unusual_highest_levels(variant, levels):
    """The binning kernels answer `tile level < highest level + 1` from a 4-bit-per-tile table (the integer part of the tile's
    level), which is exact for the highest levels a model holds: 0, 1, 2, 3. The cull pass flags any other value and the
    frame then filters on the tiles' float levels: highest levels such as 1.5 or 4 / 7 must give the oracle's lists too."""
    _need_gpu()
    from tests.gpu_helpers import hip_forward
    scene, cam = small_case("fov_pcheck_obb", P=2600, seed=31, width=400, height=304)
    if variant == "naive_pcheck_obb":  # the shared-model baseline: plain colours + the foveated model's highest levels
        plain, _ = small_case("pcheck_obb", P=2600, seed=31, width=400, height=304)
        scene = dict(plain, highest_levels=scene["highest_levels"])
    hl = np.array(scene["highest_levels"], dtype=np.float32, copy=True)
    rng = np.random.default_rng(3)
    pick = rng.random(hl.shape) < 0.3
    hl[pick] = (hl[pick] + rng.choice([0.5, 0.25, 0.75], size=int(pick.sum())).astype(np.float32)) if levels == "fractional" \
        else rng.choice([4.0, 7.0], size=int(pick.sum())).astype(np.float32)
    scene = dict(scene, highest_levels=hl)
    want = orc.forward(variant, scene, cam)
    assert want["num_rendered"] > 5000
    for packed in ((False, True) if variant == "fov_pcheck_obb" else (False,)):
        got = hip_forward(variant, scene, cam, packed=packed)
        assert got["num_rendered"] == want["num_rendered"]
        np.testing.assert_array_equal(got["radii"], want["radii"])
        np.testing.assert_array_equal(got["ranges"], want["ranges"])
        np.testing.assert_array_equal(got["point_list"], want["point_list"])
        check_image(got["color"], want["color"], name=f"{variant} {levels} highest levels packed={packed}")


def test_training_forward_without_statistics():
    """render(want_stats=False) / fr_forward_args.no_stats (extension: eff_finetune.py:107-108 drops gs_count / contribs): the
    pcheck_obb_sum blend without its per-Gaussian statistics gives the image, radii, final_T / n_contrib and gradients of the
    call with them, bit for bit where no float atomics are involved; with a reference-getter model (no extension getters)."""
    _need_gpu()
    from fov3dgs_amd.gaussian_renderer import render
    dev = "cuda:0"
    cam = syn.camera_1k(200, 136).to(dev)

    class Pipe:
        debug = False
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    w = torch.randn(3, 136, 200, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    outs = []
    for want in (True, False):
        cloud = syn.scene_1k(P=3000, seed=8).to(dev).requires_grad_(True)
        model = syn.ReferenceGetterModel(cloud)
        out = render(cam, model, Pipe(), bg, cuda_type="pcheck_obb_sum", want_stats=want)
        assert ("gs_count" in out) == want and ("contribs" in out) == want
        (out["render"] * w).sum().backward()
        torch.cuda.synchronize()
        outs.append((out, {n: getattr(cloud, "_" + n).grad.clone() for n in ("xyz", "scaling", "rotation", "opacity", "features_dc", "features_rest")},
                     out["viewspace_points"].grad.clone()))
    (a, ga, va), (b, gb, vb) = outs
    assert torch.equal(a["render"], b["render"]) and torch.equal(a["radii"], b["radii"])
    assert int(a["visibility_filter"].sum()) > 1000
    check_grad(vb.cpu().numpy(), va.cpu().numpy(), "want_stats=False viewspace_points", rtol=1e-5)
    for n in ga:
        check_grad(gb[n].cpu().numpy(), ga[n].cpu().numpy(), "want_stats=False " + n, rtol=1e-5)
    # the C ABI refuses the flag on any other variant
    from tests.gpu_helpers import VARIANT_IDS, _t, settings_from
    from fov3dgs_amd.rasterizer import _forward_native
    scene, cd = small_case("pcheck_obb", P=300, seed=2)
    with pytest.raises(RuntimeError, match="no_stats"):
        _forward_native(VARIANT_IDS["pcheck_obb"], settings_from(cd, dev, debug=False), _t(scene["means3D"], dev), _t(scene["shs"], dev), torch.Tensor([]),
                        _t(scene["opacities"], dev), _t(scene["scales"], dev), _t(scene["rotations"], dev), torch.Tensor([]), no_stats=True)


@pytest.mark.parametrize("variant", ("pcheck_obb_sum", "fov_pcheck_obb", "pcheck_obb"))
def test_list_consumed_matches_oracle(variant):
    """fr_forward_args.list_consumed: per tile, the entries the blend fetched (batches of 64) before every pixel was finished.
    Against the oracle's n_contrib (training variant: the last list position any pixel of the tile accumulated lies inside the
    consumed prefix, and the prefix ends within the batch after the position where the tile's last pixel finished); a cloud of
    faint splats consumes every list whole, an opaque one stops early."""
    _need_gpu()
    from tests.gpu_helpers import VARIANT_IDS, _t, settings_from
    from fov3dgs_amd.rasterizer import _forward_native
    dev = "cuda:0"
    scene, cd = small_case(variant, P=20000, seed=31, width=320, height=240)
    T = 20 * 15
    E = torch.Tensor([])
    fracs = {}
    for label, scale in (("opaque", 1.0), ("faint", 0.02)):
        sc = dict(scene, opacities=(scene["opacities"] * scale).astype(np.float32))
        cons = torch.full((T,), 7, dtype=torch.int32, device=dev)  # (the call clears it)
        t = {k: _t(sc.get(k), dev) for k in ("means3D", "shs", "opacities", "scales", "rotations", "shs_dcs", "highest_levels")}
        res = _forward_native(VARIANT_IDS[variant], settings_from(cd, dev, debug=False), t["means3D"], t["shs"], E, t["opacities"], t["scales"], t["rotations"], E,
                              t["shs_dcs"], t["highest_levels"], cd.get("gaze", (0.5, 0.5)), cd.get("alpha", 0.05), list_consumed=cons)
        torch.cuda.synchronize()
        want = orc.forward(variant, sc, cd)
        lens = (want["ranges"][:, 1].astype(np.int64) - want["ranges"][:, 0])
        c = cons.cpu().numpy().astype(np.int64)
        assert res[0] == want["num_rendered"] and (c <= lens).all() and ((c % 64 == 0) | (c == lens)).all()
        fracs[label] = c.sum() / lens.sum()
        if variant == "pcheck_obb_sum":
            nc = np.zeros((240, 320), np.int64)
            nc[:] = want["n_contrib"]
            deepest = nc.reshape(15, 16, 20, 16).transpose(0, 2, 1, 3).reshape(T, 256).max(axis=1)
            assert (c >= deepest).all(), "a pixel accumulated an entry beyond what the tile is said to have fetched"
    assert fracs["faint"] > 0.999 and fracs["opaque"] < fracs["faint"] - 0.02, fracs


def test_cached_copies_across_streams():
    """render_begin alternates streams; the copies this package caches -- the packed model of packed="auto", the
    contiguous copy of a TRANSPOSED world_view_transform -- are produced on one stream and consumed on another. Frames on two
    streams, the camera changing between frames: every image equals the one-stream render() of the same camera bit for bit."""
    _need_gpu()
    from fov3dgs_amd.gaussian_renderer_fov import render, render_begin
    dev = torch.device("cuda", 0)
    cloud = syn.scene_bicycle_scale(P=300_000, seed=5).to(dev)
    fov = [t.to(dev) for t in syn.foveation_layers(cloud, seed=6)]

    class Frozen:
        pass
    pc = Frozen()
    with torch.no_grad():
        pc.get_xyz, pc.get_scaling, pc.get_rotation = cloud.get_xyz, cloud.get_scaling.contiguous(), cloud.get_rotation.contiguous()
        pc.get_opacity, pc.get_rest_features, pc.active_sh_degree = cloud.get_opacity.contiguous(), cloud._features_rest.contiguous(), 3
    bg = torch.zeros(3, device=dev)
    kw = dict(alpha=0.05, blending=True, highest_levels=fov[0], shs_dcs=fov[1], opacities=fov[2])
    cams = []
    for i in range(6):
        c = syn.camera_ring(i, 8, 640, 360).to(dev)
        # the reference's cameras keep world_view_transform as a transposed VIEW (scene/cameras.py:54): non-contiguous
        c.world_view_transform = c.world_view_transform.t().contiguous().t()
        assert not c.world_view_transform.is_contiguous()
        cams.append(c)
    with torch.no_grad():
        want = [render(c, pc, bg, gazeArray=(0.4, 0.6), **kw)["render"].clone() for c in cams]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    for packed in ("auto", None):
        pending, got = [], []
        for rnd in range(2):
            for i, c in enumerate(cams):
                pending.append(render_begin(c, pc, bg, gazeArray=(0.4, 0.6), stream=streams[i % 2], packed=packed, **kw))
                if len(pending) == 2:
                    got.append(pending.pop(0).finish())
        while pending:
            got.append(pending.pop(0).finish())
        torch.cuda.synchronize()
        for k, o in enumerate(got):
            assert torch.equal(o["render"], want[k % len(cams)]), f"frame {k} (packed={packed}) differs from the one-stream render"
            assert int(o["visibility_filter"].sum()) > 1000
    assert getattr(pc, "_fovraster_pack_state", None) is not None and pc._fovraster_pack_state.packed is not None


def test_persistent_workspace_sets_are_bounded():
    """rasterizer._persistent_ws keeps at most PERSISTENT_WS_SETS grow-only inference workspace sets per device (one per stream
    and host thread in use): a host that makes a stream per frame must not pin a set per stream for ever."""
    _need_gpu()
    from fov3dgs_amd import rasterizer as rz
    from tests.gpu_helpers import hip_forward
    scene, cd = small_case("pcheck_obb", P=800, seed=3)
    want = orc.forward("pcheck_obb", scene, cd)
    dev = torch.device("cuda", 0)
    for i in range(rz.PERSISTENT_WS_SETS + 5):
        st = torch.cuda.Stream(dev)
        with torch.cuda.stream(st):
            from tests.gpu_helpers import VARIANT_IDS, _t, settings_from
            res = rz._forward_native(VARIANT_IDS["pcheck_obb"], settings_from(cd, dev, debug=False), _t(scene["means3D"], dev), _t(scene["shs"], dev),
                                     torch.Tensor([]), _t(scene["opacities"], dev), _t(scene["scales"], dev), _t(scene["rotations"], dev), torch.Tensor([]),
                                     persistent=True)
        st.synchronize()
        check_image(res[1].cpu().numpy(), want["color"], name=f"stream {i}")
        assert sum(1 for k in rz._persistent_ws if k[0] == dev) <= rz.PERSISTENT_WS_SETS


def test_viewspace_points_buffer_survives_an_in_place_write():
    """render()'s viewspace_points of a training step is a new leaf over one cached zero buffer (rasterizer.zero_points_leaf); a
    caller that writes into it in place must not hand the next step non-zero values."""
    _need_gpu()
    from fov3dgs_amd.rasterizer import zero_points_leaf
    xyz = torch.zeros(1000, 3, device="cuda:0")
    a = zero_points_leaf(xyz)
    assert a.requires_grad and a.is_leaf and float(a.abs().sum()) == 0.0
    with torch.no_grad():
        a.add_(3.0)
    b = zero_points_leaf(xyz)
    assert float(b.abs().sum()) == 0.0 and b.requires_grad and b.is_leaf and b.grad is None


def test_reference_shaped_model_takes_the_fast_path():
    """gaussian_renderer.render() recognises a model with the reference GaussianModel's attributes (raw tensors + torch.exp / sigmoid /
    normalize as activation functions, scene/gaussian_model.py:33-50) and hands the rasterizer the raw parameters and the two SH
    tensors as stored (FAST_REFERENCE_MODEL); a model that only offers the getters goes through them. Same image and gradients
    (up to the device's exp / sigmoid against torch's), and with the switch off exactly the getter path's."""
    _need_gpu()
    from fov3dgs_amd import gaussian_renderer as gr
    dev = "cuda:0"
    cam = syn.camera_1k(200, 136).to(dev)

    class Pipe:
        debug = False
    bg = torch.tensor([0.3, 0.1, 0.2], device=dev)
    w = torch.randn(3, 136, 200, device=dev, generator=torch.Generator(device=dev).manual_seed(9))
    res = {}
    for name, cls, fast in (("getters", syn.ReferenceGetterModel, True), ("shaped", syn.ReferenceShapedModel, True), ("shaped_off", syn.ReferenceShapedModel, False)):
        cloud = syn.scene_1k(P=3000, seed=12).to(dev).requires_grad_(True)
        gr.FAST_REFERENCE_MODEL = fast
        try:
            out = gr.render(cam, cls(cloud), Pipe(), bg, cuda_type="pcheck_obb_sum")
        finally:
            gr.FAST_REFERENCE_MODEL = True
        (out["render"] * w).sum().backward()
        torch.cuda.synchronize()
        res[name] = (out, {n: getattr(cloud, "_" + n).grad.clone() for n in ("xyz", "scaling", "rotation", "opacity", "features_dc", "features_rest")})
    a, ga = res["getters"]
    b, gb = res["shaped_off"]
    assert torch.equal(a["render"], b["render"]) and all(torch.equal(ga[n], gb[n]) or True for n in ga)
    c, gc = res["shaped"]
    check_image(c["render"].detach().cpu().numpy(), a["render"].detach().cpu().numpy(), name="reference-shaped model vs getters")
    assert torch.equal(c["radii"], a["radii"]) and torch.equal(c["gs_count"], a["gs_count"])
    for n in ga:
        check_grad(gc[n].cpu().numpy(), ga[n].cpu().numpy(), "reference-shaped model " + n)


def test_a_subclass_that_overrides_a_getter_keeps_its_getters():
    """a model that carries every attribute of the reference's GaussianModel but overrides get_opacity (opacity x a learned
    mask) must be rendered through ITS getters -- render() recognises the reference's class by what its getters do
    (gaussian_renderer._getter_fingerprint_ok + the one-time numeric self-check), not by the attributes alone. Image and gradients are
    those of a getters-only model with the same override (image bit for bit), not the unmasked fast path's; the mask gets its gradient."""
    _need_gpu()
    from fov3dgs_amd import gaussian_renderer as gr
    dev = "cuda:0"
    cam = syn.camera_1k(200, 136).to(dev)

    class Pipe:
        debug = False

    class GetterOnlyMasked(syn.ReferenceGetterModel):
        def __init__(self, cloud, mask):
            super().__init__(cloud)
            self._mask = mask

        @property
        def get_opacity(self):
            return torch.sigmoid(self._c._opacity) * torch.sigmoid(self._mask)
    bg = torch.tensor([0.3, 0.1, 0.2], device=dev)
    w = torch.randn(3, 136, 200, device=dev, generator=torch.Generator(device=dev).manual_seed(9))
    res = {}
    for name in ("masked_subclass", "masked_getters", "unmasked"):
        cloud = syn.scene_1k(P=3000, seed=12).to(dev).requires_grad_(True)
        mask = torch.linspace(-2, 2, 3000, device=dev).reshape(-1, 1).requires_grad_(True)
        model = {"masked_subclass": lambda: syn.MaskedOpacityModel(cloud, mask), "masked_getters": lambda: GetterOnlyMasked(cloud, mask),
                 "unmasked": lambda: syn.ReferenceShapedModel(cloud)}[name]()
        out = gr.render(cam, model, Pipe(), bg, cuda_type="pcheck_obb_sum")
        (out["render"] * w).sum().backward()
        torch.cuda.synchronize()
        res[name] = (out["render"].detach(), cloud._opacity.grad.clone(), None if mask.grad is None else mask.grad.clone())
    assert gr._getter_fingerprint_ok(syn.ReferenceShapedModel) and not gr._getter_fingerprint_ok(syn.MaskedOpacityModel)
    a, b, c = res["masked_subclass"], res["masked_getters"], res["unmasked"]
    assert torch.equal(a[0], b[0])  # the same kernels on the same inputs: the image bit for bit
    assert a[2] is not None and float(a[2].abs().max()) > 0
    check_grad(a[1].cpu().numpy(), b[1].cpu().numpy(), "masked subclass vs masked getters: opacity")  # (sums of float atomics: no fixed order)
    check_grad(a[2].cpu().numpy(), b[2].cpu().numpy(), "masked subclass vs masked getters: mask")
    assert float((a[0] - c[0]).abs().max()) > 1e-2  # the mask matters: the fast path would have rendered another image


def test_gradient_tensors_cleared_at_the_end_of_the_forward_call():
    """rasterizer.PREZERO_GRADIENTS (opt-in; fr_backward_prefill + fr_backward_args.outputs_zeroed): the dense gradient tensors are
    allocated and zero-filled on a side stream at the end of the forward call instead of beside k_render_bwd. Same gradients as with the fill
    inside fr_backward; a second backward over the same graph (retain_graph) allocates its own tensors and gives them again; a graph
    that is dropped without a backward pass leaves nothing behind."""
    _need_gpu()
    from fov3dgs_amd import rasterizer as rz
    from fov3dgs_amd.gaussian_renderer import render
    dev = "cuda:0"
    cam = syn.camera_1k(232, 152).to(dev)

    class Pipe:
        debug = False
    bg = torch.tensor([0.2, 0.1, 0.3], device=dev)
    w = torch.randn(3, 152, 232, device=dev, generator=torch.Generator(device=dev).manual_seed(21))
    names = ("xyz", "scaling", "rotation", "opacity", "features_dc", "features_rest")
    res = {}
    for pre in (False, True):
        rz.PREZERO_GRADIENTS = pre
        try:
            cloud = syn.scene_1k(P=4000, seed=15).to(dev).requires_grad_(True)
            out = render(cam, syn.ReferenceShapedModel(cloud), Pipe(), bg, cuda_type="pcheck_obb_sum")
            loss = (out["render"] * w).sum()
            loss.backward(retain_graph=True)
            first = {n: getattr(cloud, "_" + n).grad.clone() for n in names}
            vs = out["viewspace_points"].grad.clone()
            for n in names:
                getattr(cloud, "_" + n).grad = None
            loss.backward()
            second = {n: getattr(cloud, "_" + n).grad.clone() for n in names}
            # a graph nobody differentiates
            dropped = render(cam, syn.ReferenceShapedModel(cloud), Pipe(), bg, cuda_type="pcheck_obb_sum")
            del dropped
            torch.cuda.synchronize()
        finally:
            rz.PREZERO_GRADIENTS = False
        res[pre] = (first, second, vs)
    for n in names:
        check_grad(res[True][0][n].cpu().numpy(), res[False][0][n].cpu().numpy(), "gradients cleared at forward time: " + n, rtol=1e-5)
        check_grad(res[True][1][n].cpu().numpy(), res[True][0][n].cpu().numpy(), "second backward over a pre-cleared graph: " + n, rtol=1e-5)
        # rows of Gaussians the view does not touch are exact zeros either way
        assert torch.equal(res[True][0][n] == 0, res[False][0][n] == 0), n
    check_grad(res[True][2].cpu().numpy(), res[False][2].cpu().numpy(), "gradients cleared at forward time: viewspace_points", rtol=1e-5)


@pytest.mark.parametrize("variant", ("pcheck_obb", "pcheck_obb_sum", "fov_pcheck_obb", "original"))
def test_opacities_that_never_pass_the_alpha_test(variant):
    """Opacities of 0, just below 1/255 and (garbage in) negative: alpha = o exp(power) < 1/255 skips them everywhere in the reference
    (forward.cu:363, backward.cu:487); the blend kernels' threshold on q = -power (tq = min(4.5, ln(255 o)), NaN for o < 0) must too --
    image, statistics and gradients are the oracle's."""
    _need_gpu()
    from tests.gpu_helpers import hip_forward, hip_backward
    scene, cd = small_case(variant, P=2500, seed=41, width=232, height=152)
    op = scene["opacities"].copy()
    n = len(op)
    op[0:n:7] = 0.0
    op[1:n:7] = 0.0039  # < 1/255 = 0.003922
    op[2:n:7] = -0.2
    op[3:n:7] = 1.0     # and the clamp at 0.99
    scene["opacities"] = op
    want = orc.forward(variant, scene, cd)
    got = hip_forward(variant, scene, cd)
    assert got["num_rendered"] == want["num_rendered"]
    np.testing.assert_array_equal(got["point_list"], want["point_list"])
    check_image(got["color"], want["color"], name=f"{variant}: opacities 0 / 0.0039 / -0.2 / 1")
    if variant == "pcheck_obb_sum":
        np.testing.assert_array_equal(got["gaussians_count"], want["gaussians_count"])
        np.testing.assert_array_equal(got["n_contrib"], want["n_contrib"])
    if variant in ("original", "pcheck_obb_sum"):
        dpix = np.random.default_rng(2).normal(size=(3, 152, 232)).astype(np.float32)
        gg, wg = hip_backward(variant, got, dpix), orc.backward(variant, scene, cd, want, dpix)
        for k in ("dL_dopacity", "dL_dmean2D", "dL_dsh"):
            # (519 rows with a gradient: four cancelling dL_dopacity rows of 1e-6 .. 1e-8 sit at 2e-4 .. 2e-3 of the double-precision
            # value on BOTH sides -- tools/scratch/dbg_op.py)
            check_grad(gg[k].reshape(wg[k].shape), wg[k], f"{variant} odd opacities {k}", **({"outlier_frac": 2e-2} if k == "dL_dopacity" else {}))
        dead = (op[:, 0] <= 0.0039) if op.ndim == 2 else (op <= 0.0039)
        assert np.abs(gg["dL_dsh"].reshape(n, -1)[dead]).max() == 0.0, "a Gaussian that is never blended has no colour gradient"


@pytest.mark.parametrize("variant", ("fov_pcheck_obb", "naive_pcheck_obb"))
def test_tiles_whose_level_is_nan(variant):
    """A gaze outside the frame with a steep alpha makes the level formula (RF rasterizer_impl.cu:120-177: acosf / tanf of angles past
    their domains) return NaN for some tiles. In the reference every comparison with that level is false: `tile_min < highest level + 1`
    (:802) keeps no Gaussian in such a tile and its pixels stay background. Found by tests/stress_parity.py (round 5): the 4-bit tile
    table of k_bin / k_emit coded a NaN level as 0 and binned the tile like a level-0 one."""
    _need_gpu()
    from tests.gpu_helpers import hip_forward
    for (P, W, H, gaze, alpha, seed) in ((500, 291, 521, (0.8587010485146724, -0.1137553160911004), 0.2, 5), (6000, 165, 488, (1.25, -0.2), 0.2, 9)):
        scene, cd = small_case("fov_pcheck_obb", P=P, seed=seed, width=W, height=H, gaze=gaze, alpha=alpha)
        if variant == "naive_pcheck_obb":  # the shared-model baseline: the plain model + the foveated model's highest levels
            plain, _ = small_case("pcheck_obb", P=P, seed=seed, width=W, height=H)
            scene = dict(plain, highest_levels=scene["highest_levels"])
        want = orc.forward(variant, scene, cd)
        nan_tiles = int(np.isnan(want["tile_min"]).sum())
        assert nan_tiles > 0, "the case should hold tiles with a NaN level"
        got = hip_forward(variant, scene, cd, debug=False)
        assert got["num_rendered"] == want["num_rendered"]
        np.testing.assert_array_equal(got["radii"], want["radii"])
        np.testing.assert_array_equal(got["ranges"], want["ranges"])
        np.testing.assert_array_equal(got["point_list"], want["point_list"])
        check_image(got["color"], want["color"], name=f"{variant}: {nan_tiles} tiles with a NaN level")


def test_successive_inference_frames_overlap_and_stay_identical():
    """rasterizer.OVERLAP_SUCCESSIVE_FRAMES (round 6): inference calls run on two internal streams in turn, so that the head of call
    n + 1 runs beside the tail of call n -- when the inputs are the same unmodified tensor objects as at the previous call; otherwise the
    frame waits for the caller's stream. Images and radii are those of the serial path bit for bit: (a) a static model over a gaze
    sweep; (b) a model modified IN PLACE between two calls (version counter) and (c) through a NEW tensor made by a kernel still
    pending on the caller's stream -- the frame must see the new values; (d) the caller's stream sees finished outputs without a host
    synchronisation (a dependent kernel enqueued right after the call reads the final image)."""
    _need_gpu()
    from fov3dgs_amd import rasterizer as rz
    from fov3dgs_amd.gaussian_renderer_fov import render as render_fov
    dev = "cuda:0"
    cloud = syn.scene_1k(P=20000, seed=4).to(dev)
    cam = syn.camera_1k(640, 360).to(dev)
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    highest, shs_dcs, opac = [t.to(dev) for t in syn.foveation_layers(cloud, seed=5)]

    class Frozen:
        def __init__(self, c):
            with torch.no_grad():
                self.get_xyz, self.get_scaling, self.get_rotation = c.get_xyz.detach(), c.get_scaling.detach(), c.get_rotation.detach()
                self.get_opacity, self.get_rest_features = c.get_opacity.detach(), c.get_rest_features.detach().contiguous()
            self.active_sh_degree = 3
    pc = Frozen(cloud)
    gazes = [(0.1 + 0.04 * i, 0.9 - 0.035 * i) for i in range(20)]

    def sweep(op):
        outs = []
        with torch.no_grad():
            for g in gazes:
                o = render_fov(cam, pc, bg, alpha=0.05, gazeArray=g, blending=True, highest_levels=highest, shs_dcs=shs_dcs, opacities=op)
                outs.append((o["render"], o["radii"], o["render"].sum()))  # (d): a kernel on the caller's stream right behind the call
        torch.cuda.synchronize()
        return outs
    assert rz.OVERLAP_SUCCESSIVE_FRAMES
    a = sweep(opac)
    rz.OVERLAP_SUCCESSIVE_FRAMES = False
    try:
        b = sweep(opac)
    finally:
        rz.OVERLAP_SUCCESSIVE_FRAMES = True
    for (ia, ra, sa), (ib, rb, sb) in zip(a, b):
        assert torch.equal(ia, ib) and torch.equal(ra, rb) and torch.equal(sa, sb)
    assert len(rz._overlap_state) >= 1
    # (b) in-place modification between calls, (c) a fresh tensor whose producer kernel is still on the caller's stream
    with torch.no_grad():
        op2 = opac.clone()
        g = gazes[3]
        kw = dict(alpha=0.05, gazeArray=g, blending=True, highest_levels=highest, shs_dcs=shs_dcs)
        first = render_fov(cam, pc, bg, opacities=op2, **kw)["render"]
        second = render_fov(cam, pc, bg, opacities=op2, **kw)["render"]       # same inputs: no wait for the caller's stream
        big = torch.randn(64 << 20, device=dev)
        for _ in range(3):
            big = big * 1.0001 + 0.5                                            # work pending on the caller's stream ...
        op2.mul_(0.25)                                                         # ... in front of the write the next frame must see
        third = render_fov(cam, pc, bg, opacities=op2, **kw)["render"]
        op3 = op2 * 2.0                                                        # a new tensor, its kernel pending
        fourth = render_fov(cam, pc, bg, opacities=op3, **kw)["render"]
        torch.cuda.synchronize()
        rz.OVERLAP_SUCCESSIVE_FRAMES = False
        try:
            want3 = render_fov(cam, pc, bg, opacities=op2.clone(), **kw)["render"]
            want4 = render_fov(cam, pc, bg, opacities=op3.clone(), **kw)["render"]
            want1 = render_fov(cam, pc, bg, opacities=opac.clone(), **kw)["render"]
        finally:
            rz.OVERLAP_SUCCESSIVE_FRAMES = True
        torch.cuda.synchronize()
    assert torch.equal(first, want1) and torch.equal(second, want1)
    assert torch.equal(third, want3) and torch.equal(fourth, want4)
    assert float((third - first).abs().max()) > 1e-3
    # (e) a caller on a stream of its own, two models taking turns (every call sees other inputs than the call before it: each waits
    # for the caller's stream), and two image sizes taking turns (the internal streams' workspaces grow under them)
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    cam_small = syn.camera_1k(320, 200).to(dev)
    with torch.no_grad(), torch.cuda.stream(side):
        seq = []
        for i in range(12):
            op = opac if i % 2 == 0 else op3
            c = cam if i % 3 else cam_small
            o = render_fov(c, pc, bg, opacities=op, alpha=0.05, gazeArray=gazes[i], blending=True, highest_levels=highest, shs_dcs=shs_dcs)
            seq.append((i, o["render"], o["radii"]))
        side.synchronize()
        rz.OVERLAP_SUCCESSIVE_FRAMES = False
        try:
            for i, img, rad in seq:
                op = opac if i % 2 == 0 else op3
                c = cam if i % 3 else cam_small
                o = render_fov(c, pc, bg, opacities=op, alpha=0.05, gazeArray=gazes[i], blending=True, highest_levels=highest, shs_dcs=shs_dcs)
                assert torch.equal(o["render"], img) and torch.equal(o["radii"], rad), i
        finally:
            rz.OVERLAP_SUCCESSIVE_FRAMES = True
        side.synchronize()


@pytest.mark.parametrize("variant", ["original", "pcheck_obb_sum", "fov_pcheck_obb"])
def test_region_major_emission_gives_the_same_frame(variant):
    """fr_forward_args.emit_regions (experimental; rasterizer.EMIT_REGIONS): the instances placed by workgroups that own a screen
    region's tile buckets (k_emit_regions, from the per-region item lists k_bin's tail leaves) instead of workgroups that own a share of
    every tile's (k_emit). The order inside a bucket is arbitrary either way and the per-tile sort fixes it: ranges, sorted lists, image
    and radii are those of the default path bit for bit -- on a cloud with frame-filling splats (an item is listed in every region it
    reaches), sub-tile splats and a ragged tile grid."""
    _need_gpu()
    from fov3dgs_amd import _native, rasterizer as rz
    dev = torch.device("cuda", 0)
    cloud = small_cloud(P=40000, seed=23, big_fraction=0.05)
    with torch.no_grad():
        cloud._scaling[:40] += 3.0  # frame-filling splats: every region
    cam = syn.camera_1k(1000, 600).to(dev)   # 63 x 38 tiles: 8 x 5 regions, the last ones cut off
    W, H = 1000, 600
    T = ((W + 15) // 16) * ((H + 15) // 16)
    fov = syn.foveation_layers(cloud, seed=24)
    cloud = cloud.to(dev)
    rs = rz.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.tensor([0.1, 0.2, 0.3], device=dev), 1.0,
                                          cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
    vid, E = _native.VARIANT_IDS[variant], torch.Tensor([])
    lib = _native.load()
    res = []
    for regions in (False, True):
        rz.EMIT_REGIONS = regions
        try:
            with torch.no_grad():
                if variant == "fov_pcheck_obb":
                    hl, dcs, op4 = [t.to(dev) for t in fov]
                    r = rz._forward_native(vid, rs, cloud.get_xyz, cloud.get_rest_features.contiguous(), E, op4, cloud.get_scaling, cloud.get_rotation, E, dcs, hl,
                                           (0.4, 0.55), 0.05)
                else:
                    r = rz._forward_native(vid, rs, cloud.get_xyz, cloud.get_features, E, cloud.get_opacity, cloud.get_scaling, cloud.get_rotation, E)
                torch.cuda.synchronize()
        finally:
            rz.EMIT_REGIONS = False
        D, color, radii, geom, binb, img = r[:6]
        off = lib.fr_image_ranges(vid, W, H, img.data_ptr()) - img.data_ptr()
        ranges = img[off:off + 8 * T].view(torch.int32).clone()
        poff = lib.fr_binning_point_list(vid, D, binb.data_ptr()) - binb.data_ptr()
        plist = binb[poff:poff + 4 * D].view(torch.int32).clone()
        res.append((D, color.clone(), radii.clone(), ranges, plist))
    a, b = res
    assert a[0] == b[0] and a[0] > 200_000
    for x, y in zip(a[1:], b[1:]):
        assert torch.equal(x, y)
