"""Backward-only loop on the S-6M scene (one forward, K backwards): python tools/bwd_bench.py [K]. For PMC runs."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fov3dgs_amd
from fov3dgs_amd import synthetic as syn
from fov3dgs_amd.gaussian_renderer import render
K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)
cloud = syn.scene_bicycle_scale(P=6_000_000, seed=1).to(dev).requires_grad_(True)
cam = syn.camera_ring(0, 8).to(dev)
bg = torch.zeros(3, device=dev)
class Pipe: debug = False
target = torch.rand(3, cam.image_height, cam.image_width, device=dev)
o = render(cam, cloud, Pipe(), bg, cuda_type="pcheck_obb_sum")
loss = (o["render"] - target).abs().mean()
ts = []
for it in range(K):
    for p in cloud.parameters(): p.grad = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss.backward(retain_graph=True)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("bwd ms", np.round(ts, 3))
