"""Developer tool (library built with -DFR_PROJ_TIMERS): per-wave timeline of k_project on the bench frame."""
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
    if "ordered" in sys.argv[1:]:
        # the model sorted by the screen region (8 x 8 tiles) of its projected centres, the off-screen Gaussians last (tools/region_order_price.py)
        hom = torch.cat([xyz, torch.ones_like(xyz[:, :1])], 1) @ cam.full_proj_transform
        w_ = 1.0 / (hom[:, 3] + 1e-7)
        px, py = ((hom[:, 0] * w_ + 1) * W - 1) * 0.5, ((hom[:, 1] * w_ + 1) * H - 1) * 0.5
        z = xyz @ cam.world_view_transform[:3, 2] + cam.world_view_transform[3, 2]
        gxr, gyr = (W + 127) // 128, (H + 127) // 128
        reg = torch.clamp((py / 128).floor().long(), 0, gyr - 1) * gxr + torch.clamp((px / 128).floor().long(), 0, gxr - 1)
        offs = (z <= 0.2) | (px < -200) | (px > W + 200) | (py < -200) | (py > H + 200)
        perm = torch.sort(torch.where(offs, torch.full_like(reg, gxr * gyr), reg), stable=True).indices
        xyz, sc, rot, rest = xyz[perm].contiguous(), sc[perm].contiguous(), rot[perm].contiguous(), rest[perm].contiguous()
        fov = [t[perm].contiguous() for t in fov]
    rs = rz.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=dev),
                                          1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
    E = torch.Tensor([])
    for i in range(3):
        r = rz._forward_native(3, rs, xyz, rest, E, fov[2], sc, rot, E, fov[1], fov[0], (0.5, 0.5), 0.05)
    torch.cuda.synchronize()
geom = r[3]; P = xyz.shape[0]
off = ((P * 48 + 255) // 256) * 256
d = geom[off:off + 8192 * 4 * 4].view(torch.float32).view(8192, 4).cpu().numpy()
d = d[d[:, 3] > 0]
st = (d[:, 0] - d[:, 0].min()) * 10 / 1e3
first, tot = d[:, 1] * 10 / 1e3, d[:, 2] * 10 / 1e3
print("waves %d, chunks per wave %.1f" % (len(d), d[:, 3].mean()))
print("start us pct 0/50/90/100:", np.percentile(st, [0, 50, 90, 100]).round(1))
print("first chunk done after us pct 10/50/90:", np.percentile(first, [10, 50, 90]).round(1))
print("wave total us pct 10/50/90/100:", np.percentile(tot, [10, 50, 90, 100]).round(1))
print("end us pct 50/90/100:", np.percentile(st + tot, [50, 90, 100]).round(1))
print("per chunk after the first us: %.2f" % ((tot - first).mean() / (d[:, 3].mean() - 1)))

# by stretch of the cloud (tenths of the waves in index order): when its waves start, how long they take, what they end with
n = len(d)
for k in range(10):
    sl = slice(k * n // 10, (k + 1) * n // 10)
    print(f"waves {sl.start:5d}..{sl.stop:5d}: start {np.median(st[sl]):6.1f} us, wave time {np.median(tot[sl]):6.1f} (max {tot[sl].max():6.1f}), end {np.median((st + tot)[sl]):6.1f} (max {(st + tot)[sl].max():6.1f})")
