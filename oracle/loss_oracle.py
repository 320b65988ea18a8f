"""CPU restatement of the reference's image losses (TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product).

Follows fov3dgs/utils/loss_utils.py: l1_loss :17-18; gaussian / create_window :26-34 (11 taps, sigma 1.5, normalised
1-D window, 2-D window = outer product); _ssim :57-76 (five zero-padded grouped correlations, C1 = 0.01^2,
C2 = 0.03^2, mean of the map). Written with explicit shifted sums in float64 (no conv2d), gradients by autograd.
Pinned against the reference itself: tests/golden/ref_loss.npz and ref_loss2.npz were produced by importing
loss_utils.py in the build container (tests/golden/make_golden.py, make_loss_golden.py).
"""
from math import exp

import torch


def window_1d(window_size=11, sigma=1.5):
    g = torch.tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)], dtype=torch.float32)
    return (g / g.sum()).double()


def _blur(img, w2d):
    """zero-padded correlation of every channel of img [C,H,W] with the (2r+1)^2 window"""
    r = w2d.shape[0] // 2
    C, H, W = img.shape
    p = torch.zeros((C, H + 2 * r, W + 2 * r), dtype=img.dtype)
    p[:, r:r + H, r:r + W] = img
    out = torch.zeros_like(img)
    for i in range(2 * r + 1):
        for j in range(2 * r + 1):
            out = out + w2d[i, j] * p[:, i:i + H, j:j + W]
    return out


def l1_loss(a, b):
    return (a.double() - b.double()).abs().mean()


def ssim(img1, img2, window_size=11):
    x, y = img1.double(), img2.double()
    g = window_1d(window_size)
    w2d = (g[:, None].float() @ g[None, :].float()).double()  # the reference forms the 2-D window in float32
    mu1, mu2 = _blur(x, w2d), _blur(y, w2d)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = _blur(x * x, w2d) - mu1_sq
    s2 = _blur(y * y, w2d) - mu2_sq
    s12 = _blur(x * y, w2d) - mu12
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu12 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    return m.mean()


def l1_ssim(img, gt, lam):
    """-> (loss, l1, ssim, d loss / d img) with loss = (1 - lam) l1 + lam (1 - ssim)"""
    x = img.detach().double().clone().requires_grad_(True)
    l1, ss = l1_loss(x, gt), ssim(x, gt)
    loss = (1.0 - lam) * l1 + lam * (1.0 - ss)
    loss.backward()
    return loss.item(), l1.item(), ss.item(), x.grad.clone()


def ssim_with_grad(img, gt):
    x = img.detach().double().clone().requires_grad_(True)
    ss = ssim(x, gt)
    ss.backward()
    return ss.item(), x.grad.clone()
