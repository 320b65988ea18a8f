"""Fused L1+SSIM loss at 1080p: python tools/loss_bench.py [iters] (kernel times: run under tools/kstats_any.sh)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fov3dgs_amd
from fov3dgs_amd.loss_utils import l1_ssim_loss
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
a = torch.rand(3, 1080, 1920, device="cuda"); b = (a + 0.05 * torch.randn_like(a)).clamp(0, 1)
ts = []
for i in range(n):
    x = a.clone().requires_grad_(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    l1_ssim_loss(x, b, 0.2).backward()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("fused loss fwd+bwd ms: median %.3f" % np.median(ts[3:]))
