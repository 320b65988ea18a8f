import math, os, sys, torch
sys.path.insert(0, "/root/repo")
import fov3dgs_amd
from fov3dgs_amd import rasterizer as rz, synthetic as syn
dev = torch.device("cuda", 0)
cloud = syn.scene_bicycle_scale(P=6_000_000, seed=1)
fov = [t.to(dev) for t in syn.foveation_layers(cloud, seed=2)]
cloud = cloud.to(dev); E = torch.Tensor([])
W, H = int(sys.argv[1]), int(sys.argv[2])
with torch.no_grad():
    xyz, sc, rot = cloud.get_xyz, cloud.get_scaling.contiguous(), cloud.get_rotation.contiguous()
    rest = cloud.get_rest_features.contiguous()
    cam = syn.camera_ring(0, 8, width=W, height=H).to(dev)
    rs = rz.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=dev), 1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
    for i in range(130):
        rz._forward_native(3, rs, xyz, rest, E, fov[2], sc, rot, E, fov[1], fov[0], (0.5, 0.5), 0.05, persistent=True)
    torch.cuda.synchronize()
