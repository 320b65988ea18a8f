"""Developer tool: how much of each tile's depth-sorted list does the blend pass consume before every pixel of the tile
has saturated (T < 1e-4)?  Training forward of the plain variant on the bench frame; n_contrib is the last list position
a pixel used (forward.cu:333-421)."""
import math, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fov3dgs_amd  # noqa
from fov3dgs_amd import _native, rasterizer as rz, synthetic as syn
dev = torch.device("cuda", 0)
cloud = syn.scene_bicycle_scale(P=6_000_000, seed=1).to(dev)
lib = _native.load()
vid = _native.VARIANT_IDS["pcheck_obb_sum"]
for ci in (0, 3):
    cam = syn.camera_ring(ci, 8).to(dev)
    W, H = cam.image_width, cam.image_height
    TX, TY = (W + 15) // 16, (H + 15) // 16
    T = TX * TY
    rs = rz.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=dev),
                                          1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
    E = torch.Tensor([])
    with torch.no_grad():
        r = rz._forward_native(vid, rs, cloud.get_xyz, cloud._features_dc, E, cloud.get_opacity, cloud.get_scaling.contiguous(),
                               cloud.get_rotation.contiguous(), E, sh_rest=cloud._features_rest)
        torch.cuda.synchronize()
    D, img = r[0], r[5]
    def view(buf, ptr, count, dtype):
        off = ptr - buf.data_ptr()
        return buf[off:off + 4 * count].view(dtype)
    ranges = view(img, lib.fr_image_ranges(vid, W, H, img.data_ptr()), 2 * T, torch.int32).view(T, 2).long()
    nc = view(img, lib.fr_image_n_contrib(vid, W, H, img.data_ptr()), W * H, torch.int32).view(H, W).long()
    pad = torch.zeros(TY * 16, TX * 16, dtype=torch.long, device=dev); pad[:H, :W] = nc
    used = pad.view(TY, 16, TX, 16).permute(0, 2, 1, 3).reshape(T, 256).max(dim=1).values
    n = (ranges[:, 1] - ranges[:, 0])
    n_c, u_c = n.cpu().numpy().astype(np.float64), used.cpu().numpy().astype(np.float64)
    print(f"camera {ci}: D={D} tiles {T}; consumed {u_c.sum() / n_c.sum():.3f} of all list entries")
    for lo, hi in ((1, 64), (64, 512), (512, 2048), (2048, 4096), (4096, 1 << 30)):
        m = (n_c >= lo) & (n_c < hi)
        if m.any():
            print(f"  lists [{lo},{hi}): {int(m.sum()):5d} tiles, {n_c[m].sum() / n_c.sum():.3f} of the entries, consumed {u_c[m].sum() / n_c[m].sum():.3f}"
                  f" (median {np.median(u_c[m] / n_c[m]):.3f}); reach p50 {int(np.percentile(u_c[m], 50))} p90 {int(np.percentile(u_c[m], 90))} p99 {int(np.percentile(u_c[m], 99))}"
                  f" max {int(u_c[m].max())}; tiles reaching > 960: {int((u_c[m] > 960).sum())}, > 1920: {int((u_c[m] > 1920).sum())}")
