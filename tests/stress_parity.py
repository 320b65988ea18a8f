"""Randomised HIP-vs-oracle parity sweep (not collected by pytest; test_randomised_sweep is its fixed-seed sibling).
usage: python tests/stress_parity.py [rounds=24] [seed=0]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import small_cloud, small_camera, scene_dict, cam_dict, syn
from tests.gpu_helpers import hip_forward
from oracle import oracle as orc

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
variants = ("original", "pcheck_obb_sum", "pcheck_obb", "fov_pcheck_obb", "pcheck_obb_max")
bad = 0
for r in range(rounds):
    variant = variants[r % len(variants)]
    P = int(rng.choice([1, 2, 63, 64, 65, 500, 3000, 9000, 20000]))
    W, H = int(rng.integers(17, 900)), int(rng.integers(17, 600))
    big = float(rng.choice([0.0, 0.1, 0.5]))
    cloud = small_cloud(P, seed=int(rng.integers(1 << 30)), big_fraction=big) if P >= 8 else syn.scene_1k(P=P, seed=r)
    if rng.random() < 0.3 and P >= 8:
        cloud._scaling[: max(1, P // 50)] += 3.0  # a few frame-filling splats
    cam = small_camera(W, H)
    fov = syn.foveation_layers(cloud, seed=r) if variant == "fov_pcheck_obb" else None
    scene = scene_dict(cloud, variant, fov)
    cd = cam_dict(cam, gaze=(float(rng.uniform(-0.3, 1.3)), float(rng.uniform(-0.3, 1.3))), alpha=float(rng.choice([0.05, 0.02, 0.2])))
    want = orc.forward(variant, scene, cd)
    got = hip_forward(variant, scene, cd)
    ok = got["num_rendered"] == want["num_rendered"] and np.array_equal(got["radii"], want["radii"]) and \
        np.array_equal(got["ranges"], want["ranges"]) and np.array_equal(got["point_list"], want["point_list"])
    d = np.abs(got["color"] - want["color"])
    ok = ok and np.isfinite(got["color"]).all() and d.max() <= 2e-2 and np.mean(d > 1e-4) <= 2e-3
    if ok and scene.get("scales") is not None and scene.get("shs") is not None:  # packed layout: bit-identical
        pk = hip_forward(variant, scene, cd, packed=True)
        ok = pk["num_rendered"] == got["num_rendered"] and all(np.array_equal(pk[k], got[k]) for k in ("radii", "ranges", "point_list", "color"))
    print(f"{r:3d} {variant:16s} P={P:6d} {W}x{H} big={big} D={want['num_rendered']:8d} max list {int((want['ranges'][:,1]-want['ranges'][:,0]).max()):6d} "
          f"img max diff {d.max():.2e} -> {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
print("mismatches:", bad)
sys.exit(1 if bad else 0)
