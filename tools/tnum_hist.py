#!/usr/bin/env python3
"""Developer tool: distribution of the walked rectangle sizes (tiles per candidate after clipping) of the bench frames."""
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
    xyz, sc, rot, op = cloud.get_xyz, cloud.get_scaling.contiguous(), cloud.get_rotation.contiguous(), cloud.get_opacity.contiguous()
    feats, rest = cloud.get_features.contiguous(), cloud.get_rest_features.contiguous()
    rs = rz.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=dev),
                                          1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
    E = torch.Tensor([])
    for name, gaze in (("fov centre", (0.5, 0.5)), ("fov corner", (0.25, 0.25)), ("plain", None)):
        if gaze is None:
            vid = 2
            r = rz._forward_native(vid, rs, xyz, feats, E, op, sc, rot, E, persistent=True)
        else:
            vid = 3
            r = rz._forward_native(vid, rs, xyz, rest, E, fov[2], sc, rot, E, fov[1], fov[0], gaze, 0.05, persistent=True)
        torch.cuda.synchronize()
        geom = r[3]; P = xyz.shape[0]
        view = lambda ptr, count, dtype: geom[ptr - geom.data_ptr():ptr - geom.data_ptr() + 4 * count].view(dtype)
        V = int(view(lib.fr_geometry_vis_count(vid, P, geom.data_ptr()), 1, torch.int32).item())
        wr = view(lib.fr_geometry_walk_records(vid, P, geom.data_ptr()), 16 * V, torch.int32).view(V, 16)
        flags = (wr[:, 8] >> 30) & 3
        tn = torch.where((flags & 1) != 0, wr[:, 12], torch.zeros_like(wr[:, 12])).cpu().numpy().astype(np.int64)
        edges = [0, 1, 2, 3, 5, 9, 17, 33, 64, 1024, 1 << 30]
        print(f"{name}: candidates {V}, alive {int((tn > 0).sum())}, pairs walked {int(tn.sum())}, D {r[0]}")
        for a, b in zip(edges[:-1], edges[1:]):
            m = (tn >= a) & (tn < b)
            print(f"   tiles {a:>5}..{b - 1:<10} candidates {m.mean() * 100:5.1f} %   pairs {tn[m].sum() / max(tn.sum(), 1) * 100:5.1f} %")
        # slab view: per 64-candidate slab, max and sum of the small ones
        pad = (-len(tn)) % 64
        t2 = np.pad(tn, (0, pad)).reshape(-1, 64)
        for K in (4, 8, 16, 32):
            small = np.where(t2 <= K, t2, 0)
            print(f"   K={K:>2}: in-lane walks cover {small.sum() / tn.sum() * 100:5.1f} % of the pairs, mean wave-max {small.max(axis=1).mean():.1f} iterations per slab; "
                  f"balanced loop left with {((t2 > K) & (t2 < 64)).sum() and np.where((t2 > K) & (t2 < 64), t2, 0).sum(axis=1).mean() / 64:.1f} steps per slab")
