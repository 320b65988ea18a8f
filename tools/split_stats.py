"""Developer tool (library built with -DFR_SPLIT_TIMERS): per-list phase times of k_split_long on a bench frame."""
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
lib = _native.load()
with torch.no_grad():
    xyz, sc, rot = cloud.get_xyz, cloud.get_scaling.contiguous(), cloud.get_rotation.contiguous()
    rest = cloud.get_rest_features.contiguous()
    rs = rz.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=dev),
                                          1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
    E = torch.Tensor([])
    for i in range(3):
        r = rz._forward_native(3, rs, xyz, rest, E, fov[2], sc, rot, E, fov[1], fov[0], (0.5, 0.5), 0.05)
    torch.cuda.synchronize()
D, binb = r[0], r[4]
al = lambda x: (x + 255) // 256 * 256
maxc = D // 960 + D // 2048 + 16
off = al(4 * D) + al(8 * D) + al(8 * D)
ch = binb[off:off + 8 * maxc].view(torch.int32).cpu().numpy().reshape(-1, 2)
rows = []
for b in range(2000):
    d = ch.reshape(-1)[len(ch) * 2 - 8 * (b + 1):len(ch) * 2 - 8 * b]
    if d[0] < 2048 or d[0] > 20000 or d[6] <= 0:
        break
    rows.append(d.copy())
a = np.array(rows, dtype=np.float64)
print("long lists", len(a), "n mean %.0f max %.0f" % (a[:, 0].mean(), a[:, 0].max()))
names = ["load+hist", "sync", "scans", "chunk atomic", "scatter+copy", "total"]
prev = np.zeros(len(a))
for i, nm in enumerate(names):
    cur = a[:, 1 + i] * 10 / 1e3
    print("  %-13s mean %6.2f us  (cumulative %6.2f, max %6.2f)" % (nm, (cur - prev).mean(), cur.mean(), cur.max()))
    prev = cur
st = (a[:, 7] - a[:, 7].min()) * 10 / 1e3
print("start us pct 0/50/100:", np.percentile(st, [0, 50, 100]).round(1), " end max %.1f" % (st + a[:, 6] * 10 / 1e3).max())
big = np.argsort(-a[:, 0])[:3]
for i in big:
    print("   n=%d: " % a[i, 0], (a[i, 1:7] * 10 / 1e3).round(1), "start", st[i].round(1))
