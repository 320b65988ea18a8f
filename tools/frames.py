#!/usr/bin/env python3
"""Developer tool: N foveated frames of the bench protocol (S-6M, 1080p, nine gazes in turn), no events, no output -- the
plain frame loop for profilers. usage: python tools/frames.py [frames=45] [packed=0] [variant=fov]"""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fov3dgs_amd  # noqa
from fov3dgs_amd import _native, rasterizer as rz, synthetic as syn

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 45
use_packed = len(sys.argv) > 2 and sys.argv[2] == "1"
variant = sys.argv[3] if len(sys.argv) > 3 else "fov"
GAZES = [(0.25 * i, 0.25 * j) for i in range(1, 4) for j in range(1, 4)]
dev = torch.device("cuda", 0)
cloud = syn.scene_bicycle_scale(P=6_000_000, seed=1)
fov = [t.to(dev) for t in syn.foveation_layers(cloud, seed=2)]
cloud = cloud.to(dev)
cam = syn.camera_ring(0, 8).to(dev)
W, H = cam.image_width, cam.image_height
with torch.no_grad():
    xyz, sc, rot, op = cloud.get_xyz, cloud.get_scaling.contiguous(), cloud.get_rotation.contiguous(), cloud.get_opacity.contiguous()
    feats, rest = cloud.get_features.contiguous(), cloud.get_rest_features.contiguous()
    rs = rz.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=dev),
                                          1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
    E = torch.Tensor([])
    V = _native.VARIANT_IDS
    packed = rz.pack_model(xyz, sc, rot, fov[2], shs=rest, shs_dcs=fov[1], highest_levels=fov[0]) if use_packed else None

    def frame(i):
        if variant == "fov":
            return rz._forward_native(V["fov_pcheck_obb"], rs, xyz, rest, E, fov[2], sc, rot, E, fov[1], fov[0], GAZES[i % 9], 0.05, persistent=True, packed=packed)
        return rz._forward_native(V[variant], rs, xyz, feats, E, op, sc, rot, E, persistent=True)
    for i in range(9):
        frame(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(frames):
        frame(i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
print(f"{variant} packed={int(use_packed)}: {frames / el:.1f} fps ({el / frames * 1e3:.4f} ms/frame)", flush=True)
