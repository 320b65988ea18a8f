#!/usr/bin/env python3
"""Developer tool: the bench's headline protocol (S-6M, 1080p, foveated, the nine fixed gazes in turn) with per-stage times,
for the reference-tensor layout and the packed one, plus the non-foveated and training forward. One line per mode.
usage: python tools/stage9.py [frames=126] [modes=fov,packed,plain,train]"""
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fov3dgs_amd  # noqa
from fov3dgs_amd import _native, rasterizer as rz, synthetic as syn
from fov3dgs_amd.profiling import StageTimer

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 126
modes = (sys.argv[2] if len(sys.argv) > 2 else "fov,packed").split(",")
GAZES = [(0.25 * i, 0.25 * j) for i in range(1, 4) for j in range(1, 4)]
dev = torch.device("cuda", 0)
cloud = syn.scene_bicycle_scale(P=6_000_000, seed=1)
fov = [t.to(dev) for t in syn.foveation_layers(cloud, seed=2)]
cloud = cloud.to(dev)
if os.environ.get("FR_MODEL_ORDER") == "morton":
    # experiment: the model's Gaussians in Morton order of their positions (consecutive Gaussians are neighbours in space)
    with torch.no_grad():
        q = cloud._xyz
        lo, hi = q.min(0).values, q.max(0).values
        g = ((q - lo) / (hi - lo + 1e-9) * 1023.0).long().clamp(0, 1023)
        def spread(v):
            v = (v | (v << 16)) & 0x030000FF
            v = (v | (v << 8)) & 0x0300F00F
            v = (v | (v << 4)) & 0x030C30C3
            v = (v | (v << 2)) & 0x09249249
            return v
        code = spread(g[:, 0]) | (spread(g[:, 1]) << 1) | (spread(g[:, 2]) << 2)
        perm = torch.argsort(code)
        for name in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"):
            setattr(cloud, name, getattr(cloud, name)[perm].contiguous())
        fov = [t[perm].contiguous() for t in fov]
cam = syn.camera_ring(0, 8).to(dev)
W, H = cam.image_width, cam.image_height
with torch.no_grad():
    xyz, sc, rot, op = cloud.get_xyz, cloud.get_scaling.contiguous(), cloud.get_rotation.contiguous(), cloud.get_opacity.contiguous()
    feats, rest = cloud.get_features.contiguous(), cloud.get_rest_features.contiguous()
rs = rz.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=dev),
                                      1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
E = torch.Tensor([])
V = _native.VARIANT_IDS


def run(mode):
    packed = None
    with torch.no_grad():
        if mode == "packed":
            packed = rz.pack_model(xyz, sc, rot, fov[2], shs=rest, shs_dcs=fov[1], highest_levels=fov[0])

        def frame(i):
            if mode in ("fov", "packed"):
                return rz._forward_native(V["fov_pcheck_obb"], rs, xyz, rest, E, fov[2], sc, rot, E, fov[1], fov[0], GAZES[i % 9], 0.05, persistent=True, packed=packed)
            return rz._forward_native(V["pcheck_obb" if mode == "plain" else "pcheck_obb_sum"], rs, xyz, feats, E, op, sc, rot, E, persistent=True)
        for i in range(9):
            r = frame(i)
        torch.cuda.synchronize()
        best = None
        for rep in range(3):
            t = StageTimer(frames)
            t0 = time.perf_counter()
            with t:
                for i in range(frames):
                    r = frame(i)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            ms = t.stage_ms()
            t.close()
            if best is None or el < best[0]:
                best = (el, {k: round(float(np.mean([m[k] for m in ms])), 4) for k in _native.STAGES})
        # the same frames without the stage events (what they cost)
        t0 = time.perf_counter()
        for i in range(frames):
            r = frame(i)
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t0
    el, mean = best
    print(f"{mode}: {frames / el:.1f} fps ({el / frames * 1e3:.4f} ms; no events {el2 / frames * 1e3:.4f} ms) sum_stages={sum(mean.values()):.4f} "
          + " ".join(f"{k}={v:.4f}" for k, v in mean.items()), flush=True)


for m in modes:
    run(m)
