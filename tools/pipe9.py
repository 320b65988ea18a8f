#!/usr/bin/env python3
"""Developer tool: the bench's foveated frames (S-6M, 1080p, nine gazes in turn) with TWO frames in flight (render_begin /
finish on two streams) against the same frames one after the other. usage: python tools/pipe9.py [frames=126] [depth=2]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fov3dgs_amd  # noqa
from fov3dgs_amd import synthetic as syn
from fov3dgs_amd.gaussian_renderer_fov import render, render_begin

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 126
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
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
streams = [torch.cuda.Stream(dev) for _ in range(depth)]
torch.cuda.synchronize()


def sequential(n):
    with torch.no_grad():
        for i in range(n):
            out = render(cam, pc, bg, gazeArray=GAZES[i % 9], **kw)
    return out


def pipelined(n):
    pending = []
    out = None
    for i in range(n):
        pending.append(render_begin(cam, pc, bg, gazeArray=GAZES[i % 9], stream=streams[i % depth], **kw))
        if len(pending) == depth:
            out = pending.pop(0).finish()
    for p in pending:
        out = p.finish()
    return out


ref = sequential(9)["render"].clone()
got = pipelined(9)["render"]
torch.cuda.synchronize()
assert torch.equal(ref, got), "pipelined image differs"
for name, fn in (("sequential", sequential), ("pipelined", pipelined), ("sequential", sequential), ("pipelined", pipelined)):
    fn(18)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(frames)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"{name}: {frames / el:.1f} fps ({el / frames * 1e3:.4f} ms/frame)", flush=True)
