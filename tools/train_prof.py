"""Training-step (fwd+bwd) kernel breakdown on the S-6M scene: python tools/train_prof.py (under rocprofv3)."""
import math, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fov3dgs_amd
from fov3dgs_amd import synthetic as syn
from fov3dgs_amd.gaussian_renderer import render
dev = torch.device("cuda", 0)
cloud = syn.scene_bicycle_scale(P=6_000_000, seed=1).to(dev).requires_grad_(True)
cam = syn.camera_ring(0, 8).to(dev)
bg = torch.zeros(3, device=dev)
class Pipe: debug = False
target = torch.rand(3, cam.image_height, cam.image_width, device=dev)
ts = []
NIT = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for it in range(NIT):
    for p in cloud.parameters(): p.grad = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o = render(cam, cloud, Pipe(), bg, cuda_type="pcheck_obb_sum")
    torch.cuda.synchronize(); t1 = time.perf_counter()
    loss = (o["render"] - target).abs().mean()
    loss.backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3))
ts = np.array(ts[min(3, NIT - 1):])
print("per-iteration total ms:", np.round(ts.sum(1), 2))
print("fwd ms %.3f  bwd ms %.3f  total %.3f" % (np.median(ts[:, 0]), np.median(ts[:, 1]), np.median(ts.sum(1))))
