"""Image losses: the oracle against vectors made by the reference's own loss_utils.py (CPU),
and the fused HIP kernels against both (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import loss_oracle as lo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cases():
    d = np.load(os.path.join(GOLDEN, "ref_loss.npz"))
    yield "ref_loss", d["a"], d["b"], float(d["l1"]), float(d["ssim"]), float(d["loss"]), d["grad"], None
    d2 = np.load(os.path.join(GOLDEN, "ref_loss2.npz"))
    for i in range(int(d2["n"])):
        yield f"ref_loss2[{i}]", d2[f"a{i}"], d2[f"b{i}"], float(d2[f"l1_{i}"]), float(d2[f"ssim_{i}"]), float(d2[f"loss_{i}"]), \
            d2[f"grad_{i}"], d2[f"ssim_grad_{i}"]


def test_oracle_matches_reference_vectors():
    """float64 restatement vs the reference's float32 conv2d path: values to 2e-6, gradients to 1e-6 absolute
    (gradients are O(1 / (C H W)); the tolerance is ~1e-3 of their scale on the smallest case)."""
    for name, a, b, l1, ss, loss, grad, ssim_grad in _cases():
        got_loss, got_l1, got_ss, got_grad = lo.l1_ssim(torch.tensor(a), torch.tensor(b), 0.2)
        assert abs(got_l1 - l1) < 2e-6 and abs(got_ss - ss) < 2e-6 and abs(got_loss - loss) < 2e-6, name
        if name != "ref_loss2[3]":  # identical images: |x - y| has no gradient convention worth pinning at 0
            np.testing.assert_allclose(got_grad.numpy(), grad, atol=1e-6, rtol=1e-4, err_msg=name)
        if ssim_grad is not None:
            _, g = lo.ssim_with_grad(torch.tensor(a), torch.tensor(b))
            np.testing.assert_allclose(g.numpy(), ssim_grad, atol=1e-6, rtol=1e-4, err_msg=name)


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


@pytest.mark.gpu
def test_hip_losses_match_reference_vectors():
    """Tolerances: values 2e-6 absolute (fp32 separable sums vs the reference's fp32 121-tap convolutions), gradients
    1e-4 relative + 2e-7 absolute."""
    _need_gpu()
    from fov3dgs_amd import loss_utils as lu
    for name, a, b, l1, ss, loss, grad, ssim_grad in _cases():
        x = torch.tensor(a, device="cuda:0", requires_grad=True)
        y = torch.tensor(b, device="cuda:0")
        out = lu.l1_ssim_loss(x, y, 0.2)
        out.backward()
        assert abs(out.item() - loss) < 2e-6, (name, out.item(), loss)
        assert abs(lu.l1_loss(x, y).item() - l1) < 2e-6 and abs(lu.ssim(x, y).item() - ss) < 2e-6, name
        if name != "ref_loss2[3]":
            np.testing.assert_allclose(x.grad.cpu().numpy(), grad, atol=2e-7, rtol=1e-4, err_msg=name)
        if ssim_grad is not None:
            x2 = torch.tensor(a, device="cuda:0", requires_grad=True)
            (lu.ssim(x2, y) * 3.0).backward()  # upstream gradient is applied
            np.testing.assert_allclose(x2.grad.cpu().numpy(), 3.0 * ssim_grad, atol=6e-7, rtol=1e-4, err_msg=name)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 1, 1), (1, 5, 7), (3, 16, 16), (3, 17, 33), (2, 64, 48), (3, 131, 257)])
def test_hip_losses_match_oracle(shape):
    _need_gpu()
    from fov3dgs_amd import loss_utils as lu
    g = torch.Generator().manual_seed(sum(shape))
    a = torch.rand(shape, generator=g)
    b = (a + 0.1 * torch.randn(shape, generator=g)).clamp(0, 1)
    for lam in (0.2, 0.0, 1.0):
        want_loss, _, want_ss, want_grad = lo.l1_ssim(a, b, lam)
        x = a.cuda().requires_grad_(True)
        out = lu.l1_ssim_loss(x, b.cuda(), lam)
        out.backward()
        assert abs(out.item() - want_loss) < 3e-6, (shape, lam)
        np.testing.assert_allclose(x.grad.cpu().numpy(), want_grad.numpy(), atol=2e-7 + 1e-5 / a.numel(), rtol=2e-4, err_msg=str((shape, lam)))
    # the losses are differentiable in both arguments (as the reference's): gt alone, and both at once
    want_gy = lo.l1_ssim(b, a, 0.2)[3]  # symmetric loss: d/d(gt) is d/d(first argument) with the roles swapped
    y = b.cuda().requires_grad_(True)
    lu.l1_ssim_loss(a.cuda(), y, 0.2).backward()
    np.testing.assert_allclose(y.grad.cpu().numpy(), want_gy.numpy(), atol=2e-7 + 1e-5 / a.numel(), rtol=2e-4)
    x, y = a.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    lu.l1_ssim_loss(x, y, 0.2).backward()
    np.testing.assert_allclose(y.grad.cpu().numpy(), want_gy.numpy(), atol=2e-7 + 1e-5 / a.numel(), rtol=2e-4)
    np.testing.assert_allclose(x.grad.cpu().numpy(), lo.l1_ssim(a, b, 0.2)[3].numpy(), atol=2e-7 + 1e-5 / a.numel(), rtol=2e-4)
    # an upstream gradient other than 1 reaches the kernel as a device scalar
    x3 = a.cuda().requires_grad_(True)
    (lu.l1_ssim_loss(x3, b.cuda(), 0.2) * -2.5).backward()
    np.testing.assert_allclose(x3.grad.cpu().numpy(), -2.5 * x.grad.cpu().numpy(), atol=1e-9, rtol=1e-6)
    xh = a.cuda().half().requires_grad_(True)   # the gradient comes back in the input's dtype
    lu.l1_ssim_loss(xh, b.cuda(), 0.2).backward()
    assert xh.grad.dtype == torch.float16
    # [1,C,H,W] input, no gradient requested, two runs bit-identical (no float atomics)
    v1 = lu.l1_ssim_loss(a.cuda()[None], b.cuda()[None], 0.2)
    v2 = lu.l1_ssim_loss(a.cuda()[None], b.cuda()[None], 0.2)
    assert torch.equal(v1, v2) and abs(v1.item() - lo.l1_ssim(a, b, 0.2)[0]) < 3e-6


@pytest.mark.gpu
def test_hip_losses_full_size_properties():
    """1080p: ssim(x, x) = 1 with a vanishing gradient, 0 <= loss, symmetry of ssim in its arguments, and agreement
    with the oracle on a crop-independent statistic (mean over the whole frame, oracle run on the full frame)."""
    _need_gpu()
    from fov3dgs_amd import loss_utils as lu
    g = torch.Generator().manual_seed(5)
    a = torch.rand((3, 1080, 1920), generator=g, device="cpu").cuda()
    b = (a + 0.05 * torch.randn(a.shape, device="cuda")).clamp(0, 1)
    x = a.clone().requires_grad_(True)
    s = lu.ssim(x, a)
    s.backward()
    assert abs(s.item() - 1.0) < 1e-6 and float(x.grad.abs().max()) < 1e-9
    assert abs(lu.ssim(a, b).item() - lu.ssim(b, a).item()) < 1e-6
    with torch.no_grad():  # values only: the autograd graph of the 121-tap oracle would not fit at this size
        want = 0.8 * lo.l1_loss(a.cpu(), b.cpu()).item() + 0.2 * (1.0 - lo.ssim(a.cpu(), b.cpu()).item())
    assert abs(lu.l1_ssim_loss(a, b, 0.2).item() - want) < 3e-6
    with pytest.raises(RuntimeError):
        lu.ssim(a.cpu(), b.cpu())


@pytest.mark.gpu
def test_fused_activations_match_torch():
    """exp / normalize / sigmoid of the model parameters as one kernel each way: values within
    2 ulp-ish (1e-6 relative) of the torch expressions the reference's getters use, gradients 1e-5 relative."""
    _need_gpu()
    from fov3dgs_amd.activations import activate
    g = torch.Generator().manual_seed(3)
    P = 10007
    rs = (torch.randn((P, 3), generator=g) * 2 - 3).cuda().requires_grad_(True)
    rq = torch.randn((P, 4), generator=g).cuda()
    rq[:5] = 0.0                      # the clamped-denominator branch
    rq[5:10] *= 1e-14
    rq = rq.requires_grad_(True)
    ro = (torch.randn((P, 1), generator=g) * 3).cuda().requires_grad_(True)
    s, q, o = activate(rs, rq, ro)
    ws, wq, wo = torch.randn_like(s), torch.randn_like(q), torch.randn_like(o)
    (s * ws).sum().add((q * wq).sum()).add((o * wo).sum()).backward()
    got = [t.detach().cpu() for t in (s, q, o, rs.grad, rq.grad, ro.grad)]
    rs2, rq2, ro2 = (t.detach().clone().requires_grad_(True) for t in (rs, rq, ro))
    s2, q2, o2 = torch.exp(rs2), torch.nn.functional.normalize(rq2), torch.sigmoid(ro2)
    (s2 * ws).sum().add((q2 * wq).sum()).add((o2 * wo).sum()).backward()
    want = [t.detach().cpu() for t in (s2, q2, o2, rs2.grad, rq2.grad, ro2.grad)]
    for name, a, b, rtol in zip(("s", "q", "o", "ds", "dq", "do"), got, want, (1e-6, 1e-6, 1e-6, 1e-5, 1e-5, 1e-5)):
        sel = slice(10, None) if name == "dq" else slice(None)  # degenerate quaternions: value checked, gradient convention not pinned
        np.testing.assert_allclose(a[sel].numpy(), b[sel].numpy(), rtol=rtol, atol=1e-7 * float(b.abs().max()), err_msg=name)
    # missing upstream gradients count as zeros
    s3, q3, o3 = activate(rs, rq, ro)
    rs.grad = rq.grad = ro.grad = None
    s3.sum().backward()
    assert float(rq.grad.abs().max()) == 0.0 and float(ro.grad.abs().max()) == 0.0
    np.testing.assert_allclose(rs.grad.cpu().numpy(), s3.detach().cpu().numpy(), rtol=1e-6)
