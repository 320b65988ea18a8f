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
E = torch.Tensor([]); vid = 3 if len(sys.argv) < 2 else 2
with torch.no_grad():
    full, opa = cloud.get_features.contiguous(), cloud.get_opacity.contiguous()
for i in range(3):
    if vid == 3:
        r = rz._forward_native(vid, rs, xyz, rest, E, fov[2], sc, rot, E, fov[1], fov[0], syn.lissajous_gaze(20, 90), 0.05)
    else:
        r = rz._forward_native(vid, rs, xyz, full, E, opa, sc, rot, E, None, None, (0.5, 0.5), 0.05)
torch.cuda.synchronize()
geom = r[3]
# cov3D sits right after rec (3P float4, 256-aligned)
P = xyz.shape[0]
off = ((P * 48 + 255) // 256) * 256
dbg = geom[off:off + 512 * 8 * 8 * 4].view(torch.float32).view(512 * 8, 8).cpu().numpy()
dbg = dbg[:256 * 8]  # one workgroup per CU
tot, sched, load, pairs, col, pulled = [dbg[:, i] * (10 if i < 5 else 1) for i in range(6)]
bsteps = np.floor(pulled / 1024); pulled = pulled - 1024 * bsteps
bigt = dbg[:, 6] * 10
print("waves", len(tot), "wave total us: mean %.0f max %.0f" % (tot.mean() / 1e3, tot.max() / 1e3))
print("  slab loop end: mean %.1f p99 %.1f max %.1f us" % (sched.mean() / 1e3, np.percentile(sched, 99) / 1e3, sched.max() / 1e3))
for name, v in ( ("load", load), ("pairs", pairs), ("colour+write", col)):
    print("  %-14s mean %.1f us  frac %.2f" % (name, v.mean() / 1e3, v.sum() / tot.sum()))
print("slabs per wave mean %.2f max %d" % (pulled.mean(), pulled.max()))
st = np.zeros(len(tot))
print("wave start us pct 10/50/90/99:", np.percentile(st, [10, 50, 90, 99]).round(1), " workgroups starting after 20 us:", int((st.reshape(-1, 8)[:, 0] > 20).sum()))
print("end time us pct 10/50/90:", np.percentile(st + tot / 1e3, [10, 50, 90]).round(1))
print("wave total pct 50/90/99: %s" % np.percentile(tot / 1e3, [50, 90, 99]).round(0))
steps = dbg[:, 7]
print("pair steps per wave mean %.1f max %d total %.2fM pairs<=%.1fM; us/step %.2f" % (steps.mean(), steps.max(), steps.sum() / 1e6, steps.sum() * 64 / 1e6, pairs.sum() / 1e3 / steps.sum()))
w = int(np.argmax(tot))
print("slowest wave: tot %.0f sched %.0f load %.0f pairs %.0f colour %.0f us, slabs %d steps %d" % (tot[w] / 1e3, sched[w] / 1e3, load[w] / 1e3, pairs[w] / 1e3, col[w] / 1e3, pulled[w], steps[w]))


print("big walks + scan: mean %.1f us per wave (%.1f steps: %.2f us/step); balanced loop: %.1f us (%.1f steps: %.2f us/step)" % (
    bigt.mean() / 1e3, (steps - bsteps).mean(), bigt.sum() / 1e3 / max((steps - bsteps).sum(), 1), (pairs - bigt).mean() / 1e3, bsteps.mean(), (pairs - bigt).sum() / 1e3 / max(bsteps.sum(), 1)))
