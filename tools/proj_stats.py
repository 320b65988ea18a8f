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
