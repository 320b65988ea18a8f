"""How many Gaussians of the bench scene survive each culling step (near plane, empty rectangle, walk clip)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fov3dgs_amd
from fov3dgs_amd import rasterizer as rz, synthetic as syn
dev = torch.device("cuda", 0)
cloud = syn.scene_bicycle_scale(P=6_000_000, seed=1)
fov = [t.to(dev) for t in syn.foveation_layers(cloud, seed=2)]
cloud = cloud.to(dev); cam = syn.camera_ring(0, 8).to(dev)
with torch.no_grad():
    xyz = cloud.get_xyz
    h = torch.cat([xyz, torch.ones_like(xyz[:, :1])], 1)
    t = h @ cam.world_view_transform
    print("P", xyz.shape[0], "z > 0.2:", int((t[:, 2] > 0.2).sum()))
    ph = h @ cam.full_proj_transform
    ndc = ph[:, :2] / (ph[:, 3:4] + 1e-7)
    inside = (t[:, 2] > 0.2) & (ndc.abs() < 1.0).all(1)
    print("centre inside the frame:", int(inside.sum()), " within 1.3x:", int(((t[:, 2] > 0.2) & (ndc.abs() < 1.3).all(1)).sum()))
    sc, rot, rest = cloud.get_scaling.contiguous(), cloud.get_rotation.contiguous(), cloud.get_rest_features.contiguous()
    full, opa = cloud.get_features.contiguous(), cloud.get_opacity.contiguous()
W, H = cam.image_width, cam.image_height
rs = rz.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=dev),
                                      1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
E = torch.Tensor([])
r = rz._forward_native(3, rs, xyz, rest, E, fov[2], sc, rot, E, fov[1], fov[0], syn.lissajous_gaze(20, 90), 0.05)
print("foveated: visible (radii > 0)", int((r[2] > 0).sum()), "instances", r[0])
import numpy as np
g = r[3]
# slab_ctr follows rec[3P] f4, cov3D[6P] (absent for RF?), ... : search the vis count word instead
print("geom bytes", g.numel())
r = rz._forward_native(2, rs, xyz, full, E, opa, sc, rot, E, None, None, (0.5, 0.5), 0.05)
print("pcheck_obb: visible (radii > 0)", int((r[2] > 0).sum()), "instances", r[0])
r = rz._forward_native(0, rs, xyz, full, E, opa, sc, rot, E, None, None, (0.5, 0.5), 0.05)
print("original: visible (radii > 0)", int((r[2] > 0).sum()), "instances", r[0])
