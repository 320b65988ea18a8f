"""Generate the golden vectors under tests/golden/ .

Run ONLY in the build container (needs /root/reference, which does not exist on the GPU
box):  python tests/golden/make_golden.py

Part A imports the pieces of the reference that run on CPU and records their outputs on
seeded inputs (these pin the oracle to the reference):
  ref_sh.npz       fov3dgs/utils/sh_utils.py:57-113   eval_sh
  ref_camera.npz   fov3dgs/utils/graphics_utils.py:38-71 getWorld2View2 / getProjectionMatrix,
                   composed as fov3dgs/scene/cameras.py:54-57
  ref_pooling.npz  metamer/odak_perception/foveation.py:94-146 make_pooling_size_map_pixels
  ref_loss.npz     fov3dgs/utils/loss_utils.py:17-76  l1_loss / ssim
Part B freezes the oracle's own outputs on small seeded scenes (regression fixtures the GPU
parity tests also compare against):
  oracle_<variant>.npz
Only data (inputs + expected outputs) is written; no reference source is copied.
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def part_a():
    sys.path.insert(0, os.path.join(REF, "fov3dgs"))
    sys.path.insert(0, os.path.join(REF, "metamer"))
    from utils.sh_utils import eval_sh
    from utils.graphics_utils import getWorld2View2, getProjectionMatrix
    from utils.loss_utils import l1_loss, ssim
    from odak_perception.foveation import make_pooling_size_map_pixels

    rng = np.random.default_rng(1234)
    # --- SH ---
    N = 257
    sh = rng.normal(size=(N, 16, 3)).astype(np.float32)
    pos = rng.normal(size=(N, 3)).astype(np.float32) * 3
    campos = np.array([0.3, -0.2, 5.0], np.float32)
    d = torch.tensor(pos) - torch.tensor(campos)
    dirs = d / d.norm(dim=1, keepdim=True)
    out = {}
    for deg in range(4):
        out[f"rgb_deg{deg}"] = (eval_sh(deg, torch.tensor(sh).transpose(1, 2), dirs) + 0.5).numpy()
    np.savez_compressed(os.path.join(HERE, "ref_sh.npz"), sh=sh, pos=pos, campos=campos, **out)

    # --- camera matrices ---
    cams = {}
    for i in range(3):
        A = rng.normal(size=(3, 3))
        Q, _ = np.linalg.qr(A)
        if np.linalg.det(Q) < 0:
            Q[:, 0] = -Q[:, 0]
        T = rng.normal(size=3) * 2
        fovx, fovy = 0.6 + 0.3 * i, 0.5 + 0.2 * i
        wvt = torch.tensor(getWorld2View2(Q, T, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
        proj = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
        full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
        center = wvt.inverse()[3, :3]
        cams[f"R{i}"], cams[f"T{i}"] = Q, T
        cams[f"fov{i}"] = np.array([fovx, fovy])
        cams[f"wvt{i}"], cams[f"proj{i}"] = wvt.numpy(), proj.numpy()
        cams[f"full{i}"], cams[f"center{i}"] = full.numpy(), center.numpy()
    np.savez_compressed(os.path.join(HERE, "ref_camera.npz"), **cams)

    # --- pooling size at tile centres (bilinear sample of odak's per-pixel map) ---
    pool = {}
    cases = [(256, 256, (0.5, 0.5), 0.05), (1920, 1080, (0.5, 0.5), 0.05), (1920, 1080, (0.25, 0.75), 0.05),
             (1237, 822, (0.75, 0.25), 0.05)]
    for ci, (W, H, gaze, alpha) in enumerate(cases):
        m = make_pooling_size_map_pixels(list(gaze), (H, W), alpha=alpha, real_image_width=2.0,
                                         real_viewing_distance=1.0).double().numpy()
        twn, thn = (W + 15) // 16, (H + 15) // 16
        res = np.zeros((thn, twn))
        for ty in range(thn):
            for tx in range(twn):
                u = (16 * tx + 8) / W * (W - 1)
                v = (16 * ty + 8) / H * (H - 1)
                # tiles whose centre falls outside the image are extrapolated by clamping
                u0 = min(max(int(math.floor(u)), 0), W - 2)
                v0 = min(max(int(math.floor(v)), 0), H - 2)
                fu, fv = u - u0, v - v0
                res[ty, tx] = ((1 - fu) * (1 - fv) * m[v0, u0] + fu * (1 - fv) * m[v0, u0 + 1]
                               + (1 - fu) * fv * m[v0 + 1, u0] + fu * fv * m[v0 + 1, u0 + 1])
        pool[f"case{ci}"] = np.array([W, H, gaze[0], gaze[1], alpha])
        pool[f"ps{ci}"] = res
        pool[f"inside{ci}"] = np.array([[(16 * tx + 8 <= W - 1) and (16 * ty + 8 <= H - 1) for tx in range(twn)]
                                        for ty in range(thn)])
    np.savez_compressed(os.path.join(HERE, "ref_pooling.npz"), **pool)

    # --- losses ---
    a = torch.tensor(rng.random(size=(3, 48, 64)).astype(np.float32), requires_grad=True)
    b = torch.tensor(rng.random(size=(3, 48, 64)).astype(np.float32))
    l1 = l1_loss(a, b)
    ss = ssim(a, b)
    loss = 0.8 * l1 + 0.2 * (1.0 - ss)
    loss.backward()
    np.savez_compressed(os.path.join(HERE, "ref_loss.npz"), a=a.detach().numpy(), b=b.numpy(), l1=l1.item(),
                        ssim=ss.item(), loss=loss.item(), grad=a.grad.numpy())


def part_b():
    from tests.helpers import small_case
    from oracle import oracle as orc
    for variant in ("original", "pcheck_obb_sum", "pcheck_obb", "fov_pcheck_obb", "pcheck_obb_max",
                    "pcheck_obb_loss_weighted_max_count"):
        scene, cam = small_case(variant)
        o = orc.forward(variant, scene, cam)
        keep = {k: o[k] for k in ("color", "radii", "point_list", "ranges", "depths", "means2D", "conic",
                                  "tiles_touched")}
        keep["num_rendered"] = np.int64(o["num_rendered"])
        stats = variant in ("pcheck_obb_sum", "pcheck_obb_max", "pcheck_obb_loss_weighted_max_count")
        if variant == "original" or variant == "pcheck_obb_sum":
            keep["final_T"], keep["n_contrib"] = o["final_T"], o["n_contrib"]
            rng = np.random.default_rng(7)
            dpix = rng.normal(size=o["color"].shape).astype(np.float32)
            g = orc.backward(variant, scene, cam, o, dpix)
            keep["dL_dpix"] = dpix
            keep.update({"g_" + k: v for k, v in g.items()})
        if stats:
            keep["gaussians_count"], keep["contributions"] = o["gaussians_count"], o["contributions"]
            if variant != "pcheck_obb_sum":  # same image/lists as pcheck_obb_sum: keep the fixture small
                for k in ("color", "point_list", "ranges", "depths", "means2D", "conic"):
                    keep.pop(k)
        if variant == "fov_pcheck_obb":
            for k in ("tile_levels", "tile_min", "tile_blend", "tile_gx", "tile_gy", "level_ranges"):
                keep[k] = o[k]
        np.savez_compressed(os.path.join(HERE, f"oracle_{variant}.npz"), **keep)


if __name__ == "__main__":
    if os.path.isdir(REF):
        part_a()
    else:
        print("no /root/reference here: skipping part A (reference-derived vectors)")
    part_b()
    print("golden vectors written to", HERE)
