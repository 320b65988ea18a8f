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
tot, sched, load, pairs, col, pulled = [dbg[:, i] * (10 if i < 5 else 1) for i in range(6)]
print("waves", len(tot), "wave total us: mean %.0f max %.0f" % (tot.mean() / 1e3, tot.max() / 1e3))
for name, v in (("sched+barrier", sched), ("load", load), ("pairs", pairs), ("colour+write", col)):
    print("  %-14s mean %.1f us  frac %.2f" % (name, v.mean() / 1e3, v.sum() / tot.sum()))
print("slabs per wave mean %.2f max %d" % (pulled.mean(), pulled.max()))
t0 = dbg[:, 6].astype(np.int64); t1 = dbg[:, 7].astype(np.int64)
base = t0.min(); st = (t0 - base) % (1 << 24); en = (t1 - base) % (1 << 24)
print("start spread us: max %.1f ; end us: min %.1f mean %.1f max %.1f" % (st.max() / 100, en.min() / 100, en.mean() / 100, en.max() / 100))
print("wave total pct 50/90/99: %s" % np.percentile(tot / 1e3, [50, 90, 99]).round(0))
