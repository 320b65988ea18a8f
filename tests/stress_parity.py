"""Randomised HIP-vs-oracle parity sweep (not collected by pytest; test_randomised_sweep is its fixed-seed sibling).
usage: python tests/stress_parity.py [rounds=24] [seed=0]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import small_cloud, small_camera, scene_dict, cam_dict, syn
from tests.gpu_helpers import hip_forward, hip_backward
from tests.checks import grad_stats
from oracle import oracle as orc

from fov3dgs_amd import rasterizer as _rz
_rz.POISON_GRADIENTS = True  # every gradient tensor starts as NaN: an element the library fails to write shows in the comparisons
only = int(os.environ.get("STRESS_ONLY", "-1"))  # replay the sweep's random numbers but run just this round, with per-tensor detail
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
variants = ("original", "pcheck_obb_sum", "pcheck_obb", "fov_pcheck_obb", "pcheck_obb_max")
WIDE = os.environ.get("STRESS_WIDE", "0") == "1"  # also: random camera, scale modifier, SH degree, background, the shared-model variant
if WIDE:
    variants = variants + ("naive_pcheck_obb", "fov_pcheck_obb")
bad = 0
for r in range(rounds):
    variant = variants[r % len(variants)]
    P = int(rng.choice([1, 2, 63, 64, 65, 500, 3000, 9000, 20000]))
    W, H = int(rng.integers(17, 900)), int(rng.integers(17, 600))
    if os.environ.get("STRESS_BIG", "0") == "1":  # large grids (16-bit LDS histograms at 4K, global counters at 8K) and long lists (>= 16384: regrouped first)
        P = int(rng.choice([3000, 40000, 150000]))
        W, H = int(rng.choice([1920, 2560, 3840, 5000, 7680])), int(rng.choice([1080, 1440, 2160, 2800, 4320]))
    big = float(rng.choice([0.0, 0.1, 0.5]))
    cloud = small_cloud(P, seed=int(rng.integers(1 << 30)), big_fraction=big) if P >= 8 else syn.scene_1k(P=P, seed=r)
    if rng.random() < 0.3 and P >= 8:
        cloud._scaling[: max(1, P // 50)] += 3.0  # a few frame-filling splats
    cam = small_camera(W, H)
    extra_cd = {}
    if WIDE:
        import math
        eye = (float(rng.uniform(-1.5, 1.5)), float(rng.uniform(-1.0, 1.0)), float(rng.uniform(-2.0, 3.0)))
        Rm, tv = syn.look_at(eye, (float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.5, 0.5)), 4.0))
        fovx = math.radians(float(rng.uniform(35.0, 110.0)))
        cam = syn.MiniCam(Rm, tv, fovx, 2 * math.atan(H / (2 * (W / (2 * math.tan(fovx / 2))))), W, H)
        extra_cd = dict(scale_modifier=float(rng.choice([1.0, 0.4, 2.5])), sh_degree=int(rng.integers(0, 4)),
                        bg=tuple(float(x) for x in rng.uniform(0, 1, 3)))
    fov = syn.foveation_layers(cloud, seed=r) if variant in ("fov_pcheck_obb", "naive_pcheck_obb") else None
    if variant == "naive_pcheck_obb":
        scene = dict(scene_dict(cloud, "pcheck_obb"), highest_levels=fov[0].numpy())
    else:
        scene = scene_dict(cloud, variant, fov)
    # opacities scaled down in a third of the rounds each by 10 / 50: translucent clouds whose lists are consumed to the end
    osc = float(rng.choice([1.0, 0.1, 0.02]))
    scene["opacities"] = (scene["opacities"] * osc).astype(np.float32)
    cd = cam_dict(cam, gaze=(float(rng.uniform(-0.3, 1.3)), float(rng.uniform(-0.3, 1.3))), alpha=float(rng.choice([0.05, 0.02, 0.2])), **extra_cd)
    if only >= 0 and r != only:
        if variant in ("original", "pcheck_obb_sum"):
            rng.normal(size=(3, H, W))
        continue
    if os.environ.get("STRESS_ROWS"):  # keep only these Gaussians of the replayed round (a failing row on its own)
        keep = np.array([int(x) for x in os.environ["STRESS_ROWS"].split(",")])
        scene = {kk: (v[keep] if isinstance(v, np.ndarray) and v.shape[:1] == (P,) else v) for kk, v in scene.items()}
    if os.environ.get("STRESS_DUMP"):
        import pickle
        pickle.dump((variant, scene, cd), open(os.environ["STRESS_DUMP"], "wb"))
        sys.exit(0)
    want = orc.forward(variant, scene, cd)
    got = hip_forward(variant, scene, cd)
    if "tile_min" in got:
        # The level map goes through acos / tan / sqrt: the device's, glibc's and CUDA's differ in the last bits, and the filter
        # `tile_min < highest level + 1` compares with an integer. A tile whose tile_min lies within those bits of an integer (seed 51
        # round 6 with STRESS_BIG: 1.0000008 in the oracle, below 1 on the device) gains or loses every Gaussian of that level at once
        # (9902 instances there): not a defect of either side, and not comparable -- such rounds are skipped, with the tile named.
        a, b = got["tile_min"], want["tile_min"]
        straddle = np.nonzero((np.floor(a) != np.floor(b)) & ~(np.isnan(a) & np.isnan(b)))[0]
        if len(straddle) and np.nanmax(np.abs(a - b)) <= 2e-5:
            print(f"{r:3d} {variant:16s} P={P:6d} {W}x{H}: skipped, tile(s) {straddle[:4]} have a level within rounding of an integer "
                  f"(device {a[straddle[0]]:.7f}, oracle {b[straddle[0]]:.7f})", flush=True)
            continue
    ok = got["num_rendered"] == want["num_rendered"] and np.array_equal(got["radii"], want["radii"]) and \
        np.array_equal(got["ranges"], want["ranges"]) and np.array_equal(got["point_list"], want["point_list"])
    d = np.abs(got["color"] - want["color"])
    if only >= 0:
        print(f"   num_rendered {got['num_rendered']} / {want['num_rendered']}, radii equal {np.array_equal(got['radii'], want['radii'])}, ranges equal "
              f"{np.array_equal(got['ranges'], want['ranges'])}, lists equal {np.array_equal(got['point_list'], want['point_list'])}, image: max {d.max():.2e}, "
              f"share of values > 1e-4: {np.mean(d > 1e-4):.2e}, > 1e-5: {np.mean(d > 1e-5):.2e}")
        if not np.array_equal(got['radii'], want['radii']):
            rows = np.nonzero(got['radii'].reshape(-1) != want['radii'].reshape(-1))[0]
            print(f"   {len(rows)} radii differ (shape {want['radii'].shape}): rows {rows[:10]}, HIP {got['radii'].reshape(-1)[rows[:10]]}, oracle {want['radii'].reshape(-1)[rows[:10]]}")
            for i in rows[:4]:
                i = int(i) % P
                print(f"     row {i}: xyz {scene['means3D'][i]}, scale {scene['scales'][i] if scene.get('scales') is not None else None}, rot {scene['rotations'][i] if scene.get('rotations') is not None else None}, opacity {scene['opacities'].reshape(-1)[i]}")
        if not np.array_equal(got['ranges'], want['ranges']):
            tl = np.nonzero((got['ranges'] != want['ranges']).any(axis=1))[0]
            print(f"   {len(tl)} tiles' ranges differ: {tl[:12]}; HIP lengths {(got['ranges'][tl[:12], 1] - got['ranges'][tl[:12], 0])}, oracle {(want['ranges'][tl[:12], 1] - want['ranges'][tl[:12], 0])}")
        if got['num_rendered'] == want['num_rendered'] and not np.array_equal(got['point_list'], want['point_list']):
            bad_pos = np.nonzero(got['point_list'] != want['point_list'])[0]
            print(f"   first differing list positions {bad_pos[:8]} of {len(bad_pos)}; tiles: {np.searchsorted(want['ranges'][:, 0].astype(np.int64), bad_pos[:8], side='right') - 1}")
    ok = ok and np.isfinite(got["color"]).all() and d.max() <= 2e-2 and np.mean(d > 1e-4) <= 2e-3
    if ok and scene.get("scales") is not None and scene.get("shs") is not None and variant != "naive_pcheck_obb":  # packed layout: bit-identical
        pk = hip_forward(variant, scene, cd, packed=True)
        ok = pk["num_rendered"] == got["num_rendered"] and all(np.array_equal(pk[k], got[k]) for k in ("radii", "ranges", "point_list", "color"))
    gnote = ""
    if (ok or only >= 0) and variant in ("original", "pcheck_obb_sum"):
        # backward: all eight gradient tensors against the oracle, row-relative (tests/checks.py); odd / tiny lists take the
        # one-entry fold of k_render_bwd, long ones the paired fold across batch boundaries
        dpix = rng.normal(size=(3, H, W)).astype(np.float32)
        # (the loss gradient is zero on the pixels whose forward state differs between the two passes: a flipped (pixel, Gaussian) pair
        # changes its pixel's transmittance for every entry behind it -- see tests/test_full_size_parity.py)
        agree = (got["n_contrib"] == want["n_contrib"]) & ~(np.abs(got["final_T"] - want["final_T"]) > 1e-5 * np.abs(want["final_T"]) + 1e-9)
        dpix *= agree[None].astype(np.float32)
        gg, wg = hip_backward(variant, got, dpix), orc.backward(variant, scene, cd, want, dpix)
        # the yardstick: the same arithmetic in double on the fp32 forward's state (tests/checks.py check_against_noise) -- small frames
        # of big faint splats are ill-conditioned for the REFERENCE's fp32 arithmetic too (1-2 % of the rows outside 1e-4); the per-Gaussian chain
        # rule is a different (matrix) formulation here: tensor by tensor it leaves 0.2-4 x the reference's share of rows outside, hence the floor
        # (and the factor 2: seed 52 round 0 with STRESS_BIG, 4409 splats of ~32 k tiles each at 4K, has 4.8 % of the REFERENCE's dL_dcov3D rows outside
        # 1e-4 of the double-precision result and 7.5 % of this library's; relative L2 of the tensor 3.3e-5)
        w64 = {kk: (v.astype(np.float64) if isinstance(v, np.ndarray) and v.dtype == np.float32 else v) for kk, v in want.items()}
        g64 = orc.backward(variant, {kk: v.astype(np.float64) for kk, v in scene.items()}, cd, w64, dpix.astype(np.float64), dtype=np.float64)
        worst = 0.0
        for k in ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot"):
            st = grad_stats(gg[k].reshape(wg[k].shape), wg[k])
            a, b = grad_stats(gg[k].reshape(wg[k].shape), g64[k]), grad_stats(wg[k], g64[k])
            worst = max(worst, a["frac_bad"] - b["frac_bad"])
            # (the whole-tensor norm is carried by a few near-camera rows of huge, cancelling gradients: judged against the double-precision
            # result like the rows, where it exceeds 1e-4 at all. Seed 61 round 71 with STRESS_WIDE: dL_dmean3D / dL_drot / dL_dscale 5.6e-5 / 4.7e-5 /
            # 1.7e-4 from the double result here against the reference arithmetic's 1.8e-3 / 1.5e-3 / 4.7e-4, dL_dcov3D 2.9e-4 against 6.2e-5:
            # the matrix form of the chain rule is the better conditioned one for the parameters, the worse one for the intermediate)
            ok = ok and np.isfinite(gg[k]).all() and (st["rel_l2"] <= 1e-4 or a["rel_l2"] <= 6.0 * b["rel_l2"] + 1e-5) and \
                a["frac_bad"] <= 2.0 * b["frac_bad"] + max(2e-3, 10.0 / max(st["rows_with_gradient"], 1))
            if only >= 0:
                print(f"   {k}: hip vs f32 oracle bad {st['frac_bad']:.2e} p99 {st['row_rel_p99']:.1e} relL2 {st['rel_l2']:.1e} | hip vs f64 {a['frac_bad']:.2e} (relL2 {a['rel_l2']:.1e}) | f32 oracle vs f64 {b['frac_bad']:.2e} (relL2 {b['rel_l2']:.1e}) rows {st['rows_with_gradient']}")
        gnote = f" grad rows outside 1e-4 beyond the fp32 reference's own: {worst:+.1e}"
    print(f"{r:3d} {variant:16s} P={P:6d} {W}x{H} big={big} opac x{osc} D={want['num_rendered']:8d}" + gnote + f" max list {int((want['ranges'][:,1]-want['ranges'][:,0]).max()):6d} "
          f"img max diff {d.max():.2e} -> {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
print("mismatches:", bad)
sys.exit(1 if bad else 0)
