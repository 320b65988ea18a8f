#!/usr/bin/env python3
"""Developer tool: the bench's foveated frames (S-6M, 1080p, nine gazes in turn) through render(), one call at a time, with
rasterizer.OVERLAP_SUCCESSIVE_FRAMES off / on / off / on (the second `off` runs with the internal streams in existence: what their
mere presence costs the serial path), per-stage kernel times of the serial frames before and after.
usage: python tools/overlap9.py [frames=126]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fov3dgs_amd  # noqa
from fov3dgs_amd import _native, rasterizer as rz, synthetic as syn
from fov3dgs_amd.gaussian_renderer_fov import render
from fov3dgs_amd.profiling import StageTimer

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 126
GAZES = [(0.25 * i, 0.25 * j) for i in range(1, 4) for j in range(1, 4)]
dev = torch.device("cuda", 0)
cloud = syn.scene_bicycle_scale(P=6_000_000, seed=1)
fov = [t.to(dev) for t in syn.foveation_layers(cloud, seed=2)]
cloud = cloud.to(dev)
cam = syn.camera_ring(0, 8).to(dev)
bg = torch.zeros(3, device=dev)


class Frozen:
    pass


pc = Frozen()
with torch.no_grad():
    pc.get_xyz = cloud.get_xyz.detach()
    pc.get_scaling, pc.get_rotation = cloud.get_scaling.detach().contiguous(), cloud.get_rotation.detach().contiguous()
    pc.get_opacity, pc.get_rest_features = cloud.get_opacity.detach().contiguous(), cloud.get_rest_features.detach().contiguous()
    pc.active_sh_degree = cloud.active_sh_degree
kw = dict(alpha=0.05, blending=True, highest_levels=fov[0], shs_dcs=fov[1], opacities=fov[2])


def run(n):
    with torch.no_grad():
        for i in range(n):
            out = render(cam, pc, bg, gazeArray=GAZES[i % 9], **kw)
    return out


def stages(n=27):
    timer = StageTimer(n)
    with timer:
        run(n)
    torch.cuda.synchronize()
    st = timer.stage_ms()
    timer.close()
    return " ".join(f"{k}={np.mean([s[k] for s in st]) * 1e3:.0f}" for k in _native.STAGES)


if os.environ.get("OV9_BENCHLIKE"):
    # the bench's timed region, piece by piece: 63-frame repeats, an (empty) StageTimer around them
    rz.OVERLAP_SUCCESSIVE_FRAMES = True
    run(45)
    torch.cuda.synchronize()
    for label, n, use_timer in (("126 plain", 126, False), ("63 plain", 63, False), ("63 + empty StageTimer", 63, True), ("126 + empty StageTimer", 126, True)):
        vals = []
        for rep in range(5):
            torch.cuda.synchronize()
            timer = StageTimer(n, stages=()) if use_timer else None
            t0 = time.perf_counter()
            if timer is not None:
                with timer:
                    run(n)
            else:
                run(n)
            torch.cuda.synchronize()
            vals.append(n / (time.perf_counter() - t0))
            if timer is not None:
                timer.close()
        print(f"overlap=True {label}: {np.median(vals):.1f} fps [{min(vals):.1f}, {max(vals):.1f}]", flush=True)
    sys.exit(0)
for mode in (False, True, False, True):
    rz.OVERLAP_SUCCESSIVE_FRAMES = mode
    run(18)
    torch.cuda.synchronize()
    best = []
    for rep in range(3):
        t0 = time.perf_counter()
        run(frames)
        torch.cuda.synchronize()
        best.append(frames / (time.perf_counter() - t0))
    print(f"overlap={mode}: {np.median(best):.1f} fps [{min(best):.1f}, {max(best):.1f}]  stages(us): {stages()}", flush=True)
