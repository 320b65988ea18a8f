#!/usr/bin/env python3
"""Developer tool (library built with -DFR_ER_TIMERS, FOVRASTER_EMIT_REGIONS=1): per-wave timeline of k_emit_regions on a bench frame."""
import math, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fov3dgs_amd  # noqa
from fov3dgs_amd import _native, rasterizer as rz, synthetic as syn
rz.OVERLAP_SUCCESSIVE_FRAMES = False
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
E = torch.Tensor([]); vid = 3; lib = _native.load()
g = int(sys.argv[1]) if len(sys.argv) > 1 else 4
gaze = [(0.25 * i, 0.25 * j) for i in range(1, 4) for j in range(1, 4)][g]
P = xyz.shape[0]
for i in range(3):
    r = rz._forward_native(vid, rs, xyz, rest, E, fov[2], sc, rot, E, fov[1], fov[0], gaze, 0.05, persistent=True)
    torch.cuda.synchronize()
geom = r[3]
off = (P * 48 + 255) // 256 * 256  # cov3D rows follow rec[3P float4] (csrc/common.h carve_geom)
nw = 1024 * 4
t = geom[off:off + nw * 32].view(torch.float32).view(nw, 8).cpu().numpy().copy()
t[:, :5] *= 0.01
live = t[:, 5] > 0
print(f"waves {nw}, with chunks {int(live.sum())}; chunks/wave mean {t[live, 5].mean():.2f} max {t[live, 5].max():.0f}; flushes/wave {t[live, 6].mean():.2f}; region changes/wave {t[live, 7].mean():.2f}")
for name, c in (("total", 0), ("grab + region tables", 1), ("fetch wait", 2), ("pair loops", 3), ("flushes", 4)):
    v = t[live, c]
    print(f"{name:22s} us: mean {v.mean():7.2f} p50 {np.percentile(v, 50):7.2f} p90 {np.percentile(v, 90):7.2f} max {v.max():7.2f}")
