"""Image losses of a training iteration (reference: fov3dgs/utils/loss_utils.py).

`l1_loss`, `ssim` keep the reference's signatures; `l1_ssim_loss(image, gt, lambda_dssim)` is the combination
eff_finetune.py:124-125 uses, (1 - l) * l1_loss + l * (1 - ssim), as ONE forward and ONE backward HIP kernel
(csrc/loss.hip) instead of five grouped 121-tap convolutions, ~15 elementwise kernels and their autograd graph.
GPU tensors only: there is no CPU fallback (NativeLibraryError / RuntimeError otherwise).
"""
import torch

from . import _native


def l1_loss(network_output, gt):
    """loss_utils.py:17-18"""
    return torch.abs((network_output - gt)).mean()


def l2_loss(network_output, gt):
    """loss_utils.py:23-24"""
    return ((network_output - gt) ** 2).mean()


def _chw(t, name):
    if t.dim() == 4 and t.size(0) == 1:
        t = t[0]
    if t.dim() != 3:
        raise RuntimeError(f"{name} must be [C,H,W] (or [1,C,H,W]), got {tuple(t.shape)}")
    if not t.is_cuda:
        raise RuntimeError("fovraster losses need GPU tensors: there is no CPU fallback")
    return t.contiguous().float()


def _forward(img, gt, want_maps, lam=0.0):
    """-> (out3 = device tensor (loss, l1, ssim), dmaps): the per-tile sums are added up on the device (fr_l1_ssim_finish:
    double, fixed order, no float atomics -- deterministic), no scalar kernels of the host framework."""
    lib = _native.load()
    C, H, W = img.shape
    nb = lib.fr_l1_ssim_blocks(C, H, W)
    partials = torch.empty((nb, 2), dtype=torch.float32, device=img.device)
    out3 = torch.empty((3,), dtype=torch.float32, device=img.device)
    dmaps = torch.empty((3, C, H, W), dtype=torch.float32, device=img.device) if want_maps else None
    with torch.cuda.device(img.device):
        stream = torch.cuda.current_stream(img.device).cuda_stream
        rc = lib.fr_l1_ssim_forward(C, H, W, img.data_ptr(), gt.data_ptr(), dmaps.data_ptr() if want_maps else None,
                                    partials.data_ptr(), stream)
        if rc == 0:
            rc = lib.fr_l1_ssim_finish(C, H, W, partials.data_ptr(), float(lam), out3.data_ptr(), stream)
    if rc != 0:
        raise RuntimeError(f"fovraster l1_ssim_forward failed ({rc}): {_native.last_error()}")
    return out3, dmaps


def _backward(img, gt, dmaps, w_l1, w_ssim, g=None):
    """g: the upstream gradient of the loss as a device scalar (read by the kernel), or None = 1"""
    lib = _native.load()
    C, H, W = img.shape
    grad = torch.empty_like(img)
    if g is not None:
        g = g.detach().reshape(-1)[:1].to(device=img.device, dtype=torch.float32).contiguous()
    with torch.cuda.device(img.device):
        rc = lib.fr_l1_ssim_backward(C, H, W, img.data_ptr(), gt.data_ptr(), dmaps.data_ptr(), float(w_l1), float(w_ssim),
                                     None if g is None else g.data_ptr(), grad.data_ptr(),
                                     torch.cuda.current_stream(img.device).cuda_stream)
    if rc != 0:
        raise RuntimeError(f"fovraster l1_ssim_backward failed ({rc}): {_native.last_error()}")
    return grad


class _L1SSIM(torch.autograd.Function):
    """loss = (1 - lam) * mean|x - y| + lam * (1 - mean ssim_map); lam = None: returns mean ssim_map itself."""

    @staticmethod
    def forward(ctx, img, gt, lam):
        x, y = _chw(img, "image"), _chw(gt, "gt")
        if x.shape != y.shape:
            raise RuntimeError(f"image {tuple(x.shape)} and gt {tuple(y.shape)} differ")
        need_x, need_y = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        out3, dmaps = _forward(x, y, need_x, 0.0 if lam is None else lam)
        ctx.lam, ctx.shapes, ctx.dtypes = lam, (img.shape, gt.shape), (img.dtype, gt.dtype)
        # both losses are symmetric in their arguments (as the reference's l1_loss / ssim are differentiable in both):
        # the gradient w.r.t. gt is the same kernel with the roles swapped, from a second set of derivative maps
        dmaps_y = _forward(y, x, True)[1] if need_y else None
        empty = x.new_empty(0)
        ctx.save_for_backward(x, y, dmaps if need_x else empty, dmaps_y if need_y else empty)
        return out3[2] if lam is None else out3[0]  # views of the device result: (loss, l1, ssim)

    @staticmethod
    def backward(ctx, g):
        x, y, dmaps, dmaps_y = ctx.saved_tensors
        n = float(x.numel())
        w_l1, w_ssim = (0.0, 1.0 / n) if ctx.lam is None else ((1.0 - ctx.lam) / n, -ctx.lam / n)
        gx = gy = None
        if ctx.needs_input_grad[0]:
            gx = _backward(x, y, dmaps, w_l1, w_ssim, g).reshape(ctx.shapes[0]).to(ctx.dtypes[0])
        if ctx.needs_input_grad[1]:
            gy = _backward(y, x, dmaps_y, w_l1, w_ssim, g).reshape(ctx.shapes[1]).to(ctx.dtypes[1])
        return gx, gy, None


def ssim(img1, img2, window_size=11, size_average=True):
    """loss_utils.py:37-46 (+ _ssim :57-76): mean SSIM with the 11x11 Gaussian window, differentiable in both images."""
    if window_size != 11 or not size_average:
        raise RuntimeError("fovraster ssim implements the reference's default call: window_size=11, size_average=True")
    return _L1SSIM.apply(img1, img2, None)


def l1_ssim_loss(image, gt, lambda_dssim=0.2):
    """(1 - lambda) * l1_loss(image, gt) + lambda * (1 - ssim(image, gt)), eff_finetune.py:124-125, fused."""
    return _L1SSIM.apply(image, gt, float(lambda_dssim))
