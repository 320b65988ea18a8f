"""N > 1 path on CPU: world_size-2 gloo processes exercise the view sharding, the image gather and the
bucketed gradient all-reduce of fov3dgs_amd/multiview.py (the GPU run uses the same code over RCCL)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import fov3dgs_amd  # noqa: F401
    from fov3dgs_amd import multiview
    r, w, _ = multiview.init_distributed()
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    views = multiview.views_for_rank(r, w, 8)
    img = torch.full((3, 4, 5), float(rank + 1))
    _, gathered = multiview.gather_images(img, dst=0)
    work, allg = multiview.all_gather_images(img, async_op=True)
    work.wait()
    params = [torch.nn.Parameter(torch.zeros(7, 3)), torch.nn.Parameter(torch.zeros(11))]
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float((rank + 1) * (i + 1)))
    info = multiview.allreduce_gradients(params)
    assert info["mode"] == "dense" and info["bytes"] == (7 * 3 + 11) * 4
    # the same sum as one flat reduce-scatter + all-gather
    fl = [torch.nn.Parameter(torch.zeros(7, 3)), torch.nn.Parameter(torch.zeros(11)), torch.nn.Parameter(torch.zeros(5, 2, 2))]
    for i, p in enumerate(fl):
        p.grad = torch.arange(p.numel(), dtype=torch.float32).view_as(p) * (rank + 1) + i
    info_f = multiview.allreduce_gradients_flat(fl)
    assert info_f["mode"] == "flat" and info_f["bytes"] == (21 + 11 + 20) * 4
    for i, p in enumerate(fl):
        want = torch.arange(p.numel(), dtype=torch.float32).view_as(p) * sum(range(1, world + 1)) + i * world
        assert torch.equal(p.grad, want)
    # gradients that are row-sparse tensors already (the rasterizer's row_sparse extension): summed over the union of the rows
    sg = [torch.nn.Parameter(torch.zeros(30, 3)), torch.nn.Parameter(torch.zeros(30, 2, 2))]
    my_rows = torch.tensor([rank, 7, 20 + rank])
    for i, p in enumerate(sg):
        vals = torch.full((3,) + tuple(p.shape[1:]), float((rank + 1) * (i + 1)))
        p.grad = torch.sparse_coo_tensor(my_rows.unsqueeze(0), vals, p.shape)
    info_s = multiview.allreduce_gradients(sg)
    assert info_s["mode"] == "sparse_rows" and info_s["rows"] == 5 and info_s["of"] == 30
    dense0 = sg[0].grad.to_dense()
    want_s = torch.zeros(30, 3); want_s[0] = 1.0; want_s[1] = 2.0; want_s[7] = 3.0; want_s[20] = 1.0; want_s[21] = 2.0
    assert sg[0].grad.is_sparse and torch.equal(dense0, want_s) and torch.equal(sg[1].grad.to_dense()[:, 0, 0], 2 * want_s[:, 0])
    ci = multiview.comm_info()
    assert ci["backend"] == "gloo" and ci["world"] == world
    # row-sparse exchange: rank r's view touches rows {r, 5}; rows outside the union stay exactly zero everywhere
    sp = [torch.nn.Parameter(torch.zeros(40, 3)), torch.nn.Parameter(torch.zeros(40, 2, 2))]
    vis = torch.zeros(40, dtype=torch.bool)
    vis[[rank, 5]] = True
    for i, p in enumerate(sp):
        p.grad = torch.zeros_like(p)
        p.grad[vis] = float((rank + 1) * (i + 1))
    info2 = multiview.allreduce_gradients(sp, visible=vis)
    assert info2["mode"] == "rows" and info2["rows"] == 3 and info2["of"] == 40
    want0 = torch.zeros(40, 3); want0[0] = 1.0; want0[1] = 2.0; want0[5] = 3.0
    assert torch.equal(sp[0].grad, want0) and torch.equal(sp[1].grad[:, 0, 0], 2 * want0[:, 0])
    # a union that covers most rows falls back to the dense exchange
    assert multiview.allreduce_gradients(sp, visible=torch.ones(40, dtype=torch.bool))["mode"] == "dense"
    # the exchange STARTED INSIDE the backward pass (OverlappedGradientExchange): the rasterizer reports every range of rows as it
    # completes it (here the test plays the rasterizer's part: rasterizer.GRADIENT_RANGE_HOOK is what fr_backward's callback calls);
    # autograd either adopts the call's tensors as p.grad or copies them -- both end with the sum over the ranks, equal to the
    # dense exchange of the same gradients
    from fov3dgs_amd import rasterizer as rz
    Pn = 1000
    gen = torch.Generator().manual_seed(100 + rank)
    for adopt in (True, False):
        named = {"means3D": torch.nn.Parameter(torch.zeros(Pn, 3)), "opacities": torch.nn.Parameter(torch.zeros(Pn, 1)),
                 "sh": torch.nn.Parameter(torch.zeros(Pn, 1, 3)), "sh_rest": torch.nn.Parameter(torch.zeros(Pn, 15, 3)), "scales": None}
        grads = {k: (None if v is None else torch.randn(v.shape, generator=gen)) for k, v in named.items()}
        grads["rotations"] = torch.randn(Pn, 4, generator=gen)  # a tensor of the call that is nobody's parameter here: not exchanged
        mine = {k: (None if v is None else v.clone()) for k, v in grads.items()}
        ex = multiview.OverlappedGradientExchange(named, ranges=4)
        assert rz.GRADIENT_RANGE_HOOK is None
        with ex:
            assert rz.GRADIENT_RANGE_HOOK is not None and rz.GRADIENT_RANGES == 4
            bounds = [0, 224, 480, 736, Pn]
            for k in range(4):
                rz.GRADIENT_RANGE_HOOK(k, bounds[k], bounds[k + 1], grads)
            for k_, p_ in named.items():      # what autograd does when backward() returns
                if p_ is not None:
                    p_.grad = grads[k_] if adopt else grads[k_].clone()
        assert rz.GRADIENT_RANGE_HOOK is None and ex.calls == [(k, bounds[k], bounds[k + 1]) for k in range(4)]
        ref = [torch.nn.Parameter(torch.zeros_like(v)) for v in named.values() if v is not None]
        for p_, k_ in zip(ref, [k for k, v in named.items() if v is not None]):
            p_.grad = mine[k_].clone()
        multiview.allreduce_gradients(ref)
        for p_, k_ in zip(ref, [k for k, v in named.items() if v is not None]):
            assert torch.equal(named[k_].grad, p_.grad), (adopt, k_)
        assert torch.equal(grads["rotations"], mine["rotations"])
    try:
        busy = {"means3D": torch.nn.Parameter(torch.zeros(4, 3))}
        busy["means3D"].grad = torch.zeros(4, 3)
        with multiview.OverlappedGradientExchange(busy):
            pass
        raise AssertionError("a gradient that exists on entry must be refused")
    except RuntimeError:
        pass
    q.put((rank, views, None if gathered is None else [g.mean().item() for g in gathered],
           allg.mean(dim=(1, 2, 3)).tolist(), [p.grad.mean().item() for p in params]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gloo_views_gather_allreduce():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (r0, v0, g0, a0, gr0), (r1, v1, g1, a1, gr1) = res
    assert v0 == [0, 1, 2, 3] and v1 == [4, 5, 6, 7]          # contiguous view blocks, no overlap
    assert g0 == [1.0, 2.0] and g1 is None                     # rank 0 holds both frames
    assert a0 == [1.0, 2.0] == a1
    assert gr0 == [3.0, 6.0] == gr1                            # sum over ranks: (1+2)*(i+1)


def test_single_process_is_a_noop():
    from fov3dgs_amd import multiview
    assert multiview.views_for_rank(0, 1, 1) == [0]
    assert multiview.views_for_rank(3, 8, 8) == [3]
    img = torch.ones(3, 2, 2)
    work, out = multiview.gather_images(img)
    assert work is None and out[0] is img
    multiview.allreduce_gradients([torch.nn.Parameter(torch.zeros(2))])
