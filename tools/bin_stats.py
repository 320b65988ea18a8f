import math, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fov3dgs_amd
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
E = torch.Tensor([]); vid = 3
for i in range(3):
    r = rz._forward_native(vid, rs, xyz, rest, E, fov[2], sc, rot, E, fov[1], fov[0], syn.lissajous_gaze(20, 90), 0.05)
torch.cuda.synchronize()
geom = r[3]
# cov3D sits right after rec (3P float4, 256-aligned)
P = xyz.shape[0]
off = ((P * 48 + 255) // 256) * 256
dbg = geom[off:off + 768 * 8 * 8 * 4].view(torch.float32).view(768 * 8, 8).cpu().numpy()
tot, pairs, steps, slabs, mx = dbg[:, 0] * 10, dbg[:, 1] * 10, dbg[:, 2], dbg[:, 3], dbg[:, 4]
print("waves", len(tot), "wave total ns: mean %.0f p50 %.0f p99 %.0f max %.0f" % (tot.mean(), np.percentile(tot, 50), np.percentile(tot, 99), tot.max()))
print("pairs ns: mean %.0f p99 %.0f max %.0f ; frac of total %.2f" % (pairs.mean(), np.percentile(pairs, 99), pairs.max(), pairs.sum() / tot.sum()))
print("steps per wave: mean %.1f p99 %.0f max %.0f ; slabs per wave mean %.2f max %.0f ; max steps in one slab %.0f" % (steps.mean(), np.percentile(steps, 99), steps.max(), slabs.mean(), slabs.max(), mx.max()))
print("ns per step %.1f" % (pairs.sum() / max(steps.sum(), 1)))
