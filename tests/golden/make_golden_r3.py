"""More reference-derived golden vectors (round 3). Run ONLY in the build container (needs /root/reference):
    python tests/golden/make_golden_r3.py
Imports the pieces of the reference that run on a CPU and records their outputs on the BENCH's own inputs:
  ref_pooling_gazes.npz  metamer/odak_perception/foveation.py:94-146 make_pooling_size_map_pixels at 1920x1080 for the nine
                         gazes of the FPS protocol (fov3dgs/render_compose_gazes_fps.py:26) and two off-screen gazes,
                         sampled at the tile centres
  ref_sh_axes.npz        fov3dgs/utils/sh_utils.py:57-113 eval_sh for view directions on and next to the coordinate axes
                         (where the degree-2 / degree-3 terms cancel) and for the bench camera's real view directions
  ref_camera_ring.npz    fov3dgs/utils/graphics_utils.py:38-71 getWorld2View2 / getProjectionMatrix composed as
                         fov3dgs/scene/cameras.py:54-57 for the eight cameras of the bench's ring
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


def main():
    sys.path.insert(0, os.path.join(REF, "fov3dgs"))
    sys.path.insert(0, os.path.join(REF, "metamer"))
    from utils.sh_utils import eval_sh
    from utils.graphics_utils import getWorld2View2, getProjectionMatrix
    from odak_perception.foveation import make_pooling_size_map_pixels
    import fov3dgs_amd  # noqa: F401
    from fov3dgs_amd import synthetic as syn

    # --- pooling size at the tile centres, the bench's gazes ---
    W, H = 1920, 1080
    gazes = [(0.25 * i, 0.25 * j) for i in range(1, 4) for j in range(1, 4)] + [(1.4, -0.2), (-0.3, 0.5)]
    twn, thn = (W + 15) // 16, (H + 15) // 16
    tx, ty = np.meshgrid(np.arange(twn), np.arange(thn))
    u = (16 * tx + 8) / W * (W - 1)
    v = (16 * ty + 8) / H * (H - 1)
    u0 = np.clip(np.floor(u).astype(int), 0, W - 2)
    v0 = np.clip(np.floor(v).astype(int), 0, H - 2)
    fu, fv = u - u0, v - v0
    pool = {"gazes": np.array(gazes), "size": np.array([W, H]), "alpha": np.float64(0.05),
            "inside": (16 * tx + 8 <= W - 1) & (16 * ty + 8 <= H - 1)}
    for gi, gaze in enumerate(gazes):
        m = make_pooling_size_map_pixels(list(gaze), (H, W), alpha=0.05, real_image_width=2.0, real_viewing_distance=1.0).double().numpy()
        pool[f"ps{gi}"] = ((1 - fu) * (1 - fv) * m[v0, u0] + fu * (1 - fv) * m[v0, u0 + 1] + (1 - fu) * fv * m[v0 + 1, u0]
                           + fu * fv * m[v0 + 1, u0 + 1]).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "ref_pooling_gazes.npz"), **pool)

    # --- SH at directions on / next to the axes and at the bench camera's view directions ---
    rng = np.random.default_rng(4321)
    axes = []
    for ax in range(3):
        for sgn in (1.0, -1.0):
            e = np.zeros(3)
            e[ax] = sgn
            axes.append(e)
            for _ in range(8):
                axes.append(e + 1e-3 * rng.normal(size=3))  # a hair off the axis
    diag = [np.array(d, float) for d in ((1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1), (1, -1, 0), (-1, 1, 1))]
    cloud = syn.scene_bicycle_scale(P=4096, seed=1)
    cam = syn.camera_ring(0, 8)
    campos = cam.camera_center.numpy().astype(np.float32)
    pos_axes = np.array([campos + 3.0 * a / np.linalg.norm(a) for a in axes + diag], np.float32)
    pos = np.concatenate([pos_axes, cloud.get_xyz.detach().numpy()[:512].astype(np.float32)])
    sh = rng.normal(size=(len(pos), 16, 3)).astype(np.float32)
    d = torch.tensor(pos) - torch.tensor(campos)
    dirs = d / d.norm(dim=1, keepdim=True)
    out = {f"rgb_deg{deg}": (eval_sh(deg, torch.tensor(sh).transpose(1, 2), dirs) + 0.5).numpy() for deg in range(4)}
    np.savez_compressed(os.path.join(HERE, "ref_sh_axes.npz"), sh=sh, pos=pos, campos=campos, n_axes=np.int64(len(pos_axes)), **out)

    # --- the eight cameras of the bench's ring ---
    cams = {}
    for i in range(8):
        c = syn.camera_ring(i, 8)
        th = 2 * math.pi * i / 8
        R, T = syn.look_at((4.0 * math.cos(th), -1.0, 4.0 * math.sin(th)), (0.0, 0.0, 0.0))  # synthetic.camera_ring's pose
        wvt = torch.tensor(getWorld2View2(np.asarray(R), np.asarray(T), np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
        proj = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=c.FoVx, fovY=c.FoVy).transpose(0, 1)
        full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
        cams[f"R{i}"], cams[f"T{i}"], cams[f"fov{i}"] = np.asarray(R), np.asarray(T), np.array([c.FoVx, c.FoVy])
        cams[f"wvt{i}"], cams[f"proj{i}"], cams[f"full{i}"], cams[f"center{i}"] = wvt.numpy(), proj.numpy(), full.numpy(), wvt.inverse()[3, :3].numpy()
    np.savez_compressed(os.path.join(HERE, "ref_camera_ring.npz"), **cams)
    print("round-3 golden vectors written to", HERE)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("no /root/reference here: these vectors can only be made in the build container")
    main()
