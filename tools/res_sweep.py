#!/usr/bin/env python3
"""Developer tool: the foveated frame at other resolutions (stage times; sanity check against performance cliffs where a
table leaves LDS or the tile scan takes another path). usage: python tools/res_sweep.py [frames=45]"""
import math, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fov3dgs_amd  # noqa
from fov3dgs_amd import _native, rasterizer as rz, synthetic as syn
from fov3dgs_amd.profiling import StageTimer
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 45
GAZES = [(0.25 * i, 0.25 * j) for i in range(1, 4) for j in range(1, 4)]
dev = torch.device("cuda", 0)
cloud = syn.scene_bicycle_scale(P=6_000_000, seed=1)
fov = [t.to(dev) for t in syn.foveation_layers(cloud, seed=2)]
cloud = cloud.to(dev)
E = torch.Tensor([])
with torch.no_grad():
    xyz, sc, rot = cloud.get_xyz, cloud.get_scaling.contiguous(), cloud.get_rotation.contiguous()
    rest = cloud.get_rest_features.contiguous()
    for W, H in ((1280, 720), (1920, 1080), (2560, 1440), (3840, 2160)):
        cam = syn.camera_ring(0, 8, width=W, height=H).to(dev)
        rs = rz.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=dev),
                                              1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
        f = lambda i: rz._forward_native(3, rs, xyz, rest, E, fov[2], sc, rot, E, fov[1], fov[0], GAZES[i % 9], 0.05, persistent=True)
        for i in range(9):
            r = f(i)
        torch.cuda.synchronize()
        t = StageTimer(frames)
        t0 = time.perf_counter()
        with t:
            for i in range(frames):
                r = f(i)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        ms = t.stage_ms(); t.close()
        mean = {k: round(float(np.mean([m[k] for m in ms])), 4) for k in _native.STAGES}
        T = ((W + 15) // 16) * ((H + 15) // 16)
        print(f"{W}x{H} ({T} tiles, D={r[0]}): {frames / el:.1f} fps " + " ".join(f"{k}={v:.4f}" for k, v in mean.items()), flush=True)
