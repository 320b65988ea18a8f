#!/usr/bin/env python3
"""Developer tool (library built with -DFR_EMIT_TIMERS, e.g. tools/ab_build.sh x_ET preprocess.hip -- -DFR_EMIT_TIMERS and
FOVRASTER_LIB=...): per-wave timeline of k_emit on the bench frame. usage: python tools/emit_stats.py [gaze index 0..8]"""
import math, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fov3dgs_amd  # noqa
from fov3dgs_amd import _native, rasterizer as rz, synthetic as syn
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
# cov3D rows follow rec[3P float4] (csrc/common.h carve_geom): 256-byte aligned
off = (P * 48 + 255) // 256 * 256
nw = 256 * 12
t = geom[off:off + nw * 32].view(torch.float32).view(nw, 8).cpu().numpy()
t[:, [0, 1, 2, 3, 4, 6, 7]] *= 0.01  # 10-ns ticks -> us (column 5 is a count)
live = t[:, 5] > 0
print(f"waves {nw}; pair steps/wave mean {t[live, 5].mean():.1f} max {t[live, 5].max():.0f}")
for name, c in (("total to barrier", 0), ("grab+fetch+unpack+single-tile stores", 1), ("scan + owner rows", 4), ("slab loop", 2), ("pair loop part", 6), ("big splats part", 7), ("vmcnt(0) at slab end", 3)):
    v = t[live, c]
    print(f"{name:18s} us: mean {v.mean():7.2f} p50 {np.percentile(v, 50):7.2f} p90 {np.percentile(v, 90):7.2f} max {v.max():7.2f}")
print("per pair step us: %.3f" % (t[live, 6].sum() / max(t[live, 5].sum(), 1)))
wg = t[:, 0].reshape(256, 12)
print("workgroup time to barrier: slowest wave per workgroup mean %.1f max %.1f; mean over waves %.1f" % (wg.max(1).mean(), wg.max(), wg.mean()))
