"""Tile-list length distribution of the bench scene (plain pcheck_obb_sum frame and the foveated frame): how many lists
each sort class of launch_tile_sort gets."""
import math, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fov3dgs_amd
from fov3dgs_amd import _native, rasterizer as rz, synthetic as syn
dev = torch.device("cuda", 0)
lib = _native.load()
cloud = syn.scene_bicycle_scale(P=6_000_000, seed=1)
fov = [t.to(dev) for t in syn.foveation_layers(cloud, seed=2)]
cloud = cloud.to(dev); cam = syn.camera_ring(0, 8).to(dev)
W, H = cam.image_width, cam.image_height
T = ((W + 15) // 16) * ((H + 15) // 16)
with torch.no_grad():
    xyz, sc, rot = cloud.get_xyz, cloud.get_scaling.contiguous(), cloud.get_rotation.contiguous()
    rest = cloud.get_rest_features.contiguous(); full, opa = cloud.get_features.contiguous(), cloud.get_opacity.contiguous()
rs = rz.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=dev),
                                      1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
E = torch.Tensor([])
GAZES = [(0.5, 0.5)] + [(x, y) for x in (0.25, 0.5, 0.75) for y in (0.25, 0.5, 0.75) if (x, y) != (0.5, 0.5)]
for name, vid, gaze in [("pcheck_obb_sum", 1, None)] + [("fov %.2f,%.2f" % g, 3, g) for g in GAZES]:
    if vid == 3:
        r = rz._forward_native(vid, rs, xyz, rest, E, fov[2], sc, rot, E, fov[1], fov[0], gaze, 0.05)
    else:
        r = rz._forward_native(vid, rs, xyz, full, E, opa, sc, rot, E, None, None, (0.5, 0.5), 0.05)
    torch.cuda.synchronize()
    img = r[5]
    import ctypes
    p = lib.fr_image_ranges(vid, W, H, img.data_ptr())
    addr = ctypes.cast(p, ctypes.c_void_p).value if not isinstance(p, int) else p
    off = addr - img.data_ptr()
    rg = img[off:off + 8 * T].view(torch.int32).view(T, 2).cpu().numpy().astype(np.int64)
    n = rg[:, 1] - rg[:, 0]
    edges = [0, 1, 512, 1024, 2048, 4096, 8192, 16384, 1 << 30]
    h, _ = np.histogram(n, bins=edges)
    print(name, "instances", int(n.sum()), "max", int(n.max()), {"%d.." % e: int(c) for e, c in zip(edges[:-1], h)},
          "entries in class", {"%d.." % e: int(n[(n >= e) & (n < e2)].sum()) for e, e2 in zip(edges[:-1], edges[1:])})
