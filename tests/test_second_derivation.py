"""The rasterizer core's second derivation and the FMA-contraction envelope (CPU suite).

(a) tests/numpy_fov_rasterizer.py -- written from the reference's CUDA files alone (RS's OBB cull, the whole RF forward: tile
    levels / infos -> filter -> level_ranges -> per-level colours -> single- and two-level blend) -- against the C oracle
    (oracle/fovraster_oracle.c), the one restatement every GPU parity test leans on: `radii`, `tiles_touched`, `ranges`,
    `point_list`, level ranges, tile level maps and per-level colours ARRAY-EQUAL in float32, images <= 1e-6 in float64, on seeded
    scenes with multi-tile splats, box-culled pairs and two-level tiles. Two hand derivations of
    RF rasterizer_impl.cu:264-383, RF forward.cu:262-476,490-609 and RS auxiliary.h:66-154 that agree bit for bit are one
    reading of the source, not one author's slip.
(b) the oracle's third flavour (float, multiply-adds fused wherever gcc's -ffp-contract=fast fuses them: what nvcc's default
    -fmad=true may do to the reference's .cu files) against the uncontracted one on the same frames: how many radii, list
    entries and pixels contraction moves. Whole S-6M / S-6M-T frames: tools/fma_envelope.py -> tests/fma_envelope_full.json.
"""
import numpy as np
import pytest

from tests import numpy_fov_rasterizer as npr
from tests import parity_report
from tests.envelope import differences
from tests.helpers import small_case
from oracle import oracle as orc

# (P, seed, gaze, width, height): ragged and whole tile grids, gazes inside / at the edge of the frame
SCENES = [(3000, 3, (0.4, 0.55), 200, 120), (3000, 5, (0.3, 0.6), 320, 208), (2500, 11, (0.95, 0.05), 256, 256), (4000, 17, (0.5, 0.5), 304, 176)]


def _pair(variant, case, dtype):
    P, seed, gaze, w, h = case
    scene, cam = small_case(variant, P=P, seed=seed, gaze=gaze, width=w, height=h)
    if dtype == np.float64:
        scene = {k: v.astype(np.float64) for k, v in scene.items()}
    return scene, cam, orc.forward(variant, scene, cam, dtype=dtype), npr.rasterize(variant, scene, cam, dtype)


@pytest.mark.parametrize("case", SCENES, ids=lambda c: f"P{c[0]}s{c[1]}_{c[3]}x{c[4]}")
@pytest.mark.parametrize("variant", ("pcheck_obb", "pcheck_obb_sum", "fov_pcheck_obb"))
def test_numpy_twin_agrees_with_the_oracle(variant, case):
    scene, cam, o, n = _pair(variant, case, np.float32)
    vis = o["radii"] > 0
    multi = int(((o["tiles_rect"] > 1) & vis).sum())
    culled = int(o["num_rect"] - o["num_rendered"])
    assert multi > 500 and culled > 2000 and o["num_rendered"] > 10_000, (multi, culled, o["num_rendered"])
    for k in ("radii", "tiles_rect", "tiles_touched", "ranges", "point_list"):
        np.testing.assert_array_equal(n[k], o[k], err_msg=k)
    np.testing.assert_array_equal(n["keys"], o["keys"])
    if variant == "fov_pcheck_obb":
        assert o["tile_blend"].sum() >= 20 and len(np.unique(o["tile_min"].astype(int))) >= 3
        for k in ("tile_levels", "tile_gx", "tile_gy", "tile_min"):
            np.testing.assert_array_equal(n[k], o[k], err_msg=k)
        np.testing.assert_array_equal(n["tile_blend"], o["tile_blend"].astype(bool))
        np.testing.assert_array_equal(n["level_ranges"][vis], o["level_ranges"][vis])
        assert (o["level_ranges"][vis, 1] > o["level_ranges"][vis, 0]).sum() > 50  # Gaussians that need colours at two or more levels
        np.testing.assert_array_equal(n["fov_colors"][vis], o["fov_colors"][vis])     # NaN = a slot the reference never writes
    else:
        np.testing.assert_array_equal(n["rgb"], o["rgb"])
    d32 = np.abs(n["color"] - o["color"])
    assert d32.max() <= 2e-6, d32.max()   # same decisions, exp rounded from double vs glibc's expf: last-bit differences only
    if variant == "pcheck_obb_sum":
        np.testing.assert_array_equal(n["n_contrib"], o["n_contrib"])
        np.testing.assert_array_equal(n["gaussians_count"], o["gaussians_count"])   # +1 per entry of every round a live tile fetches
        assert o["gaussians_count"].sum() < o["num_rendered"]                        # ... some tiles stopped before their last round
        np.testing.assert_allclose(n["contributions"], o["contributions"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(n["final_T"], o["final_T"], rtol=0, atol=1e-6)
    # double precision: both derivations, all arithmetic in double
    scene64, cam, o64, n64 = _pair(variant, case, np.float64)
    np.testing.assert_array_equal(n64["point_list"], o64["point_list"])
    d64 = np.abs(n64["color"] - o64["color"])
    assert d64.max() <= 1e-6, d64.max()
    parity_report.record("second_derivation", f"{variant} P={case[0]} seed={case[1]} {case[3]}x{case[4]} gaze={case[2]}",
                         instances=int(o["num_rendered"]), multi_tile_splats=multi, box_or_level_culled_pairs=culled,
                         two_level_tiles=int(o["tile_blend"].sum()) if variant == "fov_pcheck_obb" else 0,
                         image_f32_max_abs=float(d32.max()), image_f64_max_abs=float(d64.max()), index_outputs="array-equal")


def test_twin_two_level_tiles_differ_from_single_level_blend():
    """The comparison above has teeth on the two-level path: blending the two-level tiles as single-level ones moves the image."""
    scene, cam = small_case("fov_pcheck_obb", P=3000, seed=5, gaze=(0.3, 0.6), width=320, height=208)
    n = npr.rasterize("fov_pcheck_obb", scene, cam, np.float32)
    ar = npr._Arith(np.float32)
    pr = npr.project(ar, scene, cam)
    lv = {k: n[k] for k in ("tile_levels", "tile_gx", "tile_gy", "tile_min")}
    lv["tile_blend"] = np.zeros_like(n["tile_blend"])
    opac4 = np.asarray(scene["opacities"], np.float32).reshape(-1, 4)
    flat = npr.blend_fov(ar, pr, cam, lv, n["point_list"], n["ranges"], opac4, np.nan_to_num(n["fov_colors"]),
                         np.asarray(scene["highest_levels"], np.float32).reshape(-1))
    assert np.abs(flat["color"] - n["color"]).max() > 1e-2


@pytest.mark.skipif(not orc.has_fma_flavour(), reason="the contracted oracle flavour needs a host with FMA3")
@pytest.mark.parametrize("variant", ("pcheck_obb_sum", "fov_pcheck_obb"))
def test_fma_contraction_envelope_small_frames(variant):
    """oracle f32 (-ffp-contract=off) against oracle f32_fma (-ffp-contract=fast -mfma) on the seeded small scenes: contraction
    moves last bits of the covariance chain, so a radius (ceil of 3 sigma), a box test or a blend threshold can flip -- rarely.
    Recorded (tests/parity_report_cpu.json); asserted: the two readings stay the same frame up to a handful of flips."""
    tot = dict(radii_differ=0, instances_in_one_only=0, positions_in_another_order=0, values_gt_1e4=0, instances_a=0, gaussians=0, values=0)
    for case in SCENES:
        P, seed, gaze, w, h = case
        scene, cam = small_case(variant, P=P, seed=seed, gaze=gaze, width=w, height=h)
        a = orc.forward(variant, scene, cam)
        b = orc.forward(variant, scene, cam, fma=True)
        d = differences(a, b)
        parity_report.record("fma_envelope", f"{variant} P={P} seed={seed} {w}x{h}: f32 vs f32_fma", **d)
        for k in tot:
            tot[k] += d[k]
        assert np.abs(a["conic"] - b["conic"]).max() > 0  # the flavours do differ in the last bits
    assert tot["radii_differ"] <= 1e-3 * tot["gaussians"], tot
    assert tot["instances_in_one_only"] <= 1e-3 * tot["instances_a"], tot
    assert tot["values_gt_1e4"] <= 1e-4 * tot["values"], tot


def test_numpy_twin_random_sweep():
    """Thirty random small frames -- cloud size, seed, image shape (ragged tile grids down to 3 x 3 tiles), gaze inside and outside
    the frame, alpha 0.02 / 0.05 / 0.2, SH degree 0..3, scale modifier -- through both derivations: radii, tile counts, ranges,
    sorted lists (and, foveated, tile_min and the level ranges) array-equal in float32, images within last-bit rounding."""
    rng = np.random.default_rng(2026)
    for it in range(30):
        variant = ("pcheck_obb", "pcheck_obb_sum", "fov_pcheck_obb")[it % 3]
        P, seed = int(rng.integers(200, 3000)), int(rng.integers(0, 10000))
        w, h = int(rng.integers(40, 400)), int(rng.integers(40, 300))
        gaze = (float(rng.uniform(-0.2, 1.2)), float(rng.uniform(-0.2, 1.2)))
        alpha = float(rng.choice([0.02, 0.05, 0.2]))
        scene, cam = small_case(variant, P=P, seed=seed, gaze=gaze, alpha=alpha, width=w, height=h)
        cam["sh_degree"] = int(rng.integers(0, 4))
        cam["scale_modifier"] = float(rng.choice([1.0, 0.5, 1.7]))
        o, n = orc.forward(variant, scene, cam), npr.rasterize(variant, scene, cam, np.float32)
        tag = f"case {it}: {variant} P={P} seed={seed} {w}x{h} gaze={gaze} alpha={alpha} deg={cam['sh_degree']} mod={cam['scale_modifier']}"
        for k in ("radii", "tiles_touched", "ranges", "point_list"):
            np.testing.assert_array_equal(n[k], o[k], err_msg=tag + " " + k)
        if variant == "fov_pcheck_obb":
            vis = o["radii"] > 0
            np.testing.assert_array_equal(n["tile_min"], o["tile_min"], err_msg=tag)
            np.testing.assert_array_equal(n["level_ranges"][vis], o["level_ranges"][vis], err_msg=tag)
        assert np.abs(n["color"] - o["color"]).max() <= 5e-6, tag
