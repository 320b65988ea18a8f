#!/usr/bin/env python3
"""Developer tool: per-gaze stage times of the foveated bench frame (reference tensors). usage: python tools/stage9_gaze.py [frames per gaze=30]"""
import math, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fov3dgs_amd  # noqa
from fov3dgs_amd import _native, rasterizer as rz, synthetic as syn
from fov3dgs_amd.profiling import StageTimer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
GAZES = [(0.25 * i, 0.25 * j) for i in range(1, 4) for j in range(1, 4)]
dev = torch.device("cuda", 0)
cloud = syn.scene_bicycle_scale(P=6_000_000, seed=1)
fov = [t.to(dev) for t in syn.foveation_layers(cloud, seed=2)]
cloud = cloud.to(dev); cam = syn.camera_ring(0, 8).to(dev)
W, H = cam.image_width, cam.image_height
with torch.no_grad():
    xyz, sc, rot = cloud.get_xyz, cloud.get_scaling.contiguous(), cloud.get_rotation.contiguous()
    rest = cloud.get_rest_features.contiguous()
    rs = rz.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=dev),
                                          1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
    E = torch.Tensor([])
    tot = {k: 0.0 for k in _native.STAGES}
    for g in GAZES:
        f = lambda: rz._forward_native(3, rs, xyz, rest, E, fov[2], sc, rot, E, fov[1], fov[0], g, 0.05, persistent=True)
        for i in range(3):
            r = f()
        torch.cuda.synchronize()
        t = StageTimer(n)
        with t:
            for i in range(n):
                r = f()
        torch.cuda.synchronize()
        ms = t.stage_ms(); t.close()
        m = {k: float(np.mean([x[k] for x in ms])) for k in _native.STAGES}
        for k in m: tot[k] += m[k] / 9
        print(f"gaze {g}: D={r[0]:>8} sum={sum(m.values()):.3f} " + " ".join(f"{k}={v:.3f}" for k, v in m.items()), flush=True)
    print(f"mean: sum={sum(tot.values()):.4f} " + " ".join(f"{k}={v:.4f}" for k, v in tot.items()))
