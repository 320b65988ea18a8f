"""Golden vectors for the image losses from the REFERENCE's own code (run in the build container only):
    python tests/golden/make_loss_golden.py   ->  tests/golden/ref_loss2.npz
imports /root/reference/fov3dgs/utils/loss_utils.py (l1_loss :17-18, ssim :37-76) on CPU and stores inputs, values and
autograd gradients for image sizes that are not multiples of the kernels' 16x16 tile."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/fov3dgs")
from utils.loss_utils import l1_loss, ssim  # noqa: E402

rng = np.random.default_rng(2025)
out = {}
cases = [(3, 37, 53, "noise"), (3, 70, 33, "smooth"), (1, 9, 5, "noise"), (3, 16, 16, "equal")]
for ci, (C, H, W, kind) in enumerate(cases):
    if kind == "smooth":  # a smooth picture and a slightly degraded copy, like a render next to its ground truth
        yy, xx = np.meshgrid(np.linspace(0, 3, H), np.linspace(0, 2, W), indexing="ij")
        base = np.stack([0.5 + 0.4 * np.sin(xx * (c + 1) + yy) * np.cos(yy * 1.7 - c) for c in range(C)]).astype(np.float32)
        a_np = np.clip(base + 0.03 * rng.standard_normal(base.shape).astype(np.float32), 0, 1)
        b_np = base
    elif kind == "equal":
        a_np = rng.random((C, H, W)).astype(np.float32); b_np = a_np.copy()
    else:
        a_np = rng.random((C, H, W)).astype(np.float32); b_np = rng.random((C, H, W)).astype(np.float32)
    a = torch.tensor(a_np, requires_grad=True); b = torch.tensor(b_np)
    l1, ss = l1_loss(a, b), ssim(a, b)
    loss = 0.8 * l1 + 0.2 * (1.0 - ss)
    loss.backward()
    a2 = torch.tensor(a_np, requires_grad=True)
    ssim(a2, b).backward()
    out.update({f"a{ci}": a_np, f"b{ci}": b_np, f"l1_{ci}": l1.item(), f"ssim_{ci}": ss.item(), f"loss_{ci}": loss.item(),
                f"grad_{ci}": a.grad.numpy(), f"ssim_grad_{ci}": a2.grad.numpy()})
out["n"] = len(cases)
np.savez_compressed(os.path.join(HERE, "ref_loss2.npz"), **out)
print("wrote ref_loss2.npz", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k.startswith(("l1_", "ssim_"))and not k.startswith("ssim_grad")})
