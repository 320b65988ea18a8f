#!/usr/bin/env python3
"""Per-stage kernel times of one rasterizer variant on the S-6M scene (developer tool).
usage: python tools/stage_bench.py [variant=fov_pcheck_obb] [frames=60, the gaze path of bench.py] [points=6000000]"""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fov3dgs_amd  # noqa
from fov3dgs_amd import _native, rasterizer as rz, synthetic as syn
from fov3dgs_amd.profiling import StageTimer

variant = sys.argv[1] if len(sys.argv) > 1 else "fov_pcheck_obb"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 60
P = int(sys.argv[3]) if len(sys.argv) > 3 else 6_000_000
dev = torch.device("cuda", 0)
cloud = syn.scene_bicycle_scale(P=P, seed=1)
fov = [t.to(dev) for t in syn.foveation_layers(cloud, seed=2)]
cloud = cloud.to(dev)
cam = syn.camera_ring(0, 8).to(dev)
W, H = cam.image_width, cam.image_height
with torch.no_grad():
    xyz, sc, rot, op = cloud.get_xyz, cloud.get_scaling.contiguous(), cloud.get_rotation.contiguous(), cloud.get_opacity.contiguous()
    feats, rest = cloud.get_features.contiguous(), cloud.get_rest_features.contiguous()
rs = rz.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=dev),
                                      1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
vid = _native.VARIANT_IDS[variant]
E = torch.Tensor([])
packed = None
if os.environ.get("FR_PACKED"):  # the static-model layout (include/fovraster.h packed_geom / packed_colour)
    with torch.no_grad():
        if variant == "fov_pcheck_obb":
            packed = rz.pack_model(xyz, sc, rot, fov[2], shs=rest, shs_dcs=fov[1], highest_levels=fov[0])
        else:
            packed = rz.pack_model(xyz, sc, rot, op, shs=feats)


def frame(i):
    if variant == "fov_pcheck_obb":
        return rz._forward_native(vid, rs, xyz, rest, E, fov[2], sc, rot, E, fov[1], fov[0], syn.lissajous_gaze(i, 90), 0.05, persistent=True, packed=packed)
    return rz._forward_native(vid, rs, xyz, feats, E, op, sc, rot, E, persistent=True, packed=packed)


for i in range(5):
    r = frame(i)
torch.cuda.synchronize()
t = StageTimer(frames)
with t:
    for i in range(frames):
        r = frame(10 + i)
torch.cuda.synchronize()
ms = t.stage_ms()
mean = {k: round(float(np.mean([m[k] for m in ms])), 4) for k in _native.STAGES}
print(variant, "D=%d" % r[0], "sum=%.3f" % sum(mean.values()), mean, flush=True)
