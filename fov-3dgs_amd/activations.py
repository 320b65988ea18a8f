"""exp / normalize / sigmoid of a 3DGS model's raw parameters as one HIP kernel each way (csrc/activations.hip).

Reference: GaussianModel.get_scaling / get_rotation / get_opacity (fov3dgs/scene/gaussian_model.py:200-240), three
torch expressions whose ~18 forward + backward kernels cost 0.45 ms per training iteration at 6 M Gaussians.
A model opts in by exposing `get_activated` (see synthetic.GaussianCloud); render() then uses it instead of the three
getters. GPU tensors only (no CPU fallback)."""
import torch

from . import _native


class _Activate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw_scaling, raw_rotation, raw_opacity):
        lib = _native.load()
        rs, rq, ro = (t.detach().contiguous().float() for t in (raw_scaling, raw_rotation, raw_opacity))
        if not rs.is_cuda:
            raise RuntimeError("fovraster activations need GPU tensors: there is no CPU fallback")
        P = rs.shape[0]
        if tuple(rs.shape) != (P, 3) or tuple(rq.shape) != (P, 4) or ro.numel() != P:
            raise RuntimeError(f"expected [P,3], [P,4], [P,1], got {tuple(rs.shape)}, {tuple(rq.shape)}, {tuple(ro.shape)}")
        s, q, o = torch.empty_like(rs), torch.empty_like(rq), torch.empty_like(ro)
        with torch.cuda.device(rs.device):
            rc = lib.fr_activate_forward(P, rs.data_ptr(), rq.data_ptr(), ro.data_ptr(), s.data_ptr(), q.data_ptr(), o.data_ptr(),
                                         torch.cuda.current_stream(rs.device).cuda_stream)
        if rc != 0:
            raise RuntimeError(f"fovraster activate_forward failed ({rc}): {_native.last_error()}")
        ctx.save_for_backward(rs, rq, ro)
        return s, q, o

    @staticmethod
    def backward(ctx, gs, gq, go):
        lib = _native.load()
        rs, rq, ro = ctx.saved_tensors
        P = rs.shape[0]
        gs, gq, go = (None if g is None else g.contiguous().float() for g in (gs, gq, go))
        ds, dq, do = torch.empty_like(rs), torch.empty_like(rq), torch.empty_like(ro)
        ptr = lambda t: None if t is None else t.data_ptr()
        with torch.cuda.device(rs.device):
            rc = lib.fr_activate_backward(P, rs.data_ptr(), rq.data_ptr(), ro.data_ptr(), ptr(gs), ptr(gq), ptr(go), ds.data_ptr(),
                                          dq.data_ptr(), do.data_ptr(), torch.cuda.current_stream(rs.device).cuda_stream)
        if rc != 0:
            raise RuntimeError(f"fovraster activate_backward failed ({rc}): {_native.last_error()}")
        return ds, dq, do


def activate(raw_scaling, raw_rotation, raw_opacity):
    """-> (exp(raw_scaling), normalize(raw_rotation), sigmoid(raw_opacity)), differentiable."""
    return _Activate.apply(raw_scaling, raw_rotation, raw_opacity)
