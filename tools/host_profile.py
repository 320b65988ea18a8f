#!/usr/bin/env python3
"""Developer tool: where the HOST's time goes in a foveated render() call (cProfile over 300 bench frames, one call at a time).
usage: python tools/host_profile.py"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fov3dgs_amd  # noqa
from fov3dgs_amd import synthetic as syn
from fov3dgs_amd.gaussian_renderer_fov import render

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
            render(cam, pc, bg, gazeArray=GAZES[i % 9], **kw)


run(27)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(300)
torch.cuda.synchronize()
print(f"{300 / (time.perf_counter() - t0):.1f} fps unprofiled")
pr = cProfile.Profile()
pr.enable()
run(300)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
