"""Candidates of the cull pass vs visible Gaussians on the bench gaze path: python tools/cand_stats.py (developer tool)."""
import math, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fov3dgs_amd
from fov3dgs_amd import rasterizer as rz, synthetic as syn
dev = torch.device("cuda", 0)
P = 6_000_000
cloud = syn.scene_bicycle_scale(P=P, seed=1)
fov = [t.to(dev) for t in syn.foveation_layers(cloud, seed=2)]
cloud = cloud.to(dev); cam = syn.camera_ring(0, 8).to(dev)
with torch.no_grad():
    xyz, sc, rot, rest = cloud.get_xyz, cloud.get_scaling.contiguous(), cloud.get_rotation.contiguous(), cloud.get_rest_features.contiguous()
W, H = cam.image_width, cam.image_height
rs = rz.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=dev),
                                      1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
E = torch.Tensor([])
al = lambda o: (o + 255) // 256 * 256
off = al(P * 48); off = al(off + P * 24); off = al(off + P * 64); off = al(off + P * 64); off = al(off + P * 4)  # rec, cov3D, wrec, lvl, lrange
for i in (10, 25, 40, 55):
    r = rz._forward_native(3, rs, xyz, rest, E, fov[2], sc, rot, E, fov[1], fov[0], syn.lissajous_gaze(i, 90), 0.05, persistent=True)
    torch.cuda.synchronize()
    ctr = r[3][off:off + 8].view(torch.int32)
    print("frame", i, "candidates", int(ctr[1]), "visible", int((r[2] > 0).sum()), "instances", r[0])
