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
E = torch.Tensor([]); vid = 3; lib = _native.load()
for i in range(3):
    r = rz._forward_native(vid, rs, xyz, rest, E, fov[2], sc, rot, E, fov[1], fov[0], syn.lissajous_gaze(20, 90), 0.05)
torch.cuda.synchronize()
img = r[5]; T = ((W + 15) // 16) * ((H + 15) // 16)
def view(ptr, n, dt):
    off = ptr - img.data_ptr(); return img[off:off + 4 * n].view(dt)
ft = view(lib.fr_image_final_T(vid, W, H, img.data_ptr()), 2 * T, torch.float32).cpu().numpy()
nc2 = view(lib.fr_image_n_contrib(vid, W, H, img.data_ptr()), 4 * T, torch.int32).cpu().numpy()
nc, nh, tloop, tsync = nc2[:T], nc2[T:2 * T], nc2[2 * T:3 * T] * 10.0, nc2[3 * T:] * 10.0
rg = view(lib.fr_image_ranges(vid, W, H, img.data_ptr()), 2 * T, torch.int32).cpu().numpy().reshape(T, 2)
n = rg[:, 1] - rg[:, 0]
cyc = ft[:T] * 10.0  # ns (100 MHz)
start = ft[T:]
print("tiles", T, "sum list", n.sum(), "processed", nc.sum(), "with a hit", nh.sum())
print("wave time ns: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (cyc.mean(), np.percentile(cyc, 50), np.percentile(cyc, 90), np.percentile(cyc, 99), cyc.max()))
order = np.argsort(-cyc)[:8]
for t in order: print("tile", t, "n", n[t], "processed", nc[t], "hit", nh[t], "loop ns", tloop[t], "topsync ns", tsync[t], "ns", cyc[t], "ns/entry %.1f" % (cyc[t] / max(nc[t], 1)), "start", start[t])
st = (start - start.min()) % (1 << 24)
print("start spread ns: p50 %.0f p99 %.0f max %.0f" % (np.percentile(st, 50) * 10, np.percentile(st, 99) * 10, st.max() * 10))
print("sum wave time ms", cyc.sum() / 1e6, " / 4096 slots =", cyc.sum() / 4096 / 1e6)
end = st * 10 + cyc
print("end ns: p50 %.0f p90 %.0f p99 %.0f max %.0f" % tuple(np.percentile(end, [50, 90, 99, 100])))
late = np.argsort(-end)[:8]
for t in late: print("late tile", t, "n", n[t], "processed", nc[t], "hit", nh[t], "start", st[t] * 10, "dur", cyc[t])

lv = view(lib.fr_image_tile_levels(W, H, img.data_ptr()), 5 * T, torch.float32).cpu().numpy().reshape(5, T)
bl = lv[4] != 0
print("blend tiles", int(bl.sum()), "of", T, " wave time ns mean: blend %.0f single %.0f; ns/processed entry: blend %.1f single %.1f" % (
    cyc[bl].mean(), cyc[~bl].mean(), cyc[bl].sum() / max(nc[bl].sum(), 1), cyc[~bl].sum() / max(nc[~bl].sum(), 1)))
print("slowest 20 tiles: blend among them", int(bl[np.argsort(-cyc)[:20]].sum()), " longest list among blend", n[bl].max(), "single", n[~bl].max())
print("sum wave time: blend %.2f ms single %.2f ms" % (cyc[bl].sum() / 1e6, cyc[~bl].sum() / 1e6))
