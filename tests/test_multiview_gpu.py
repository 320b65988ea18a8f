"""Config 5 at world size 2 on ONE GPU: two ranks share cuda:0 (so the collective runs over gloo -- RCCL refuses two
ranks on a device), each renders its own camera of the ring with the training variant, takes the fused loss, runs the
backward pass and sums the gradients with multiview.allreduce_gradients. Every rank then repeats BOTH views alone and
checks that the exchanged gradients are the sum of the two per-view gradients."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from tests.helpers import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Pipe:
    debug = False


def _view_grads(syn, render, l1_ssim_loss, cloud_cpu, view, dev, W, H):
    cloud = cloud_cpu.to(dev).requires_grad_(True)
    cam = syn.camera_ring(view, 8, W, H).to(dev)
    target = torch.rand(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(100 + view))
    out = render(cam, cloud, _Pipe(), torch.zeros(3, device=dev), cuda_type="pcheck_obb_sum")
    l1_ssim_loss(out["render"], target, 0.2).backward()
    return cloud, out["visibility_filter"]


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world))
        import torch.distributed as dist
        import fov3dgs_amd  # noqa: F401
        from fov3dgs_amd import multiview, synthetic as syn
        from fov3dgs_amd.gaussian_renderer import render
        from fov3dgs_amd.loss_utils import l1_ssim_loss
        r, w, local = multiview.init_distributed()
        shared = world > torch.cuda.device_count()
        assert (r, w) == (rank, world) and dist.get_backend() == ("gloo" if shared else "nccl")
        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)
        W, H = 320, 192
        cloud_cpu = syn.scene_bicycle_scale(P=60_000, seed=1, scale_log_mean=-3.2)
        mine, vis = _view_grads(syn, render, l1_ssim_loss, cloud_cpu, rank, dev, W, H)
        params = mine.parameters()
        own = [p.grad.clone() for p in params]
        res = {}
        for mode, kw in (("dense", {}), ("rows", dict(visible=vis, sparse_below=1.1))):
            for p, g in zip(params, own):
                p.grad = g.clone()
            info = multiview.allreduce_gradients(params, **kw)
            assert info["mode"] == mode and info["bytes"] > 0, info
            res[mode] = [p.grad.clone() for p in params]
        # the other rank's view, alone, on this rank
        other, _ = _view_grads(syn, render, l1_ssim_loss, cloud_cpu, 1 - rank, dev, W, H)
        worst = 0.0
        for i, (g_own, p_other) in enumerate(zip(own, other.parameters())):
            want = g_own + p_other.grad
            scale = float(want.abs().max()) + 1e-12
            for mode in ("dense", "rows"):
                err = float((res[mode][i] - want).abs().max()) / scale
                worst = max(worst, err)
                # float atomics make a view's gradient sums order-dependent in the last bits; nothing else differs
                assert err < 2e-5, (mode, i, err)
            assert float(want.abs().max()) > 0
        # the same sum STARTED INSIDE the backward pass (multiview.OverlappedGradientExchange; round 6): a model whose rasterizer inputs
        # are its leaf parameters (raw activations, split SH), the per-Gaussian pass in four ranges of rows, every range's all-reduce on
        # a communication stream as soon as the range is complete
        def fused_view(view, exchange):
            cloud = cloud_cpu.to(dev).requires_grad_(True)
            cloud.fuse_activations = True
            cam = syn.camera_ring(view, 8, W, H).to(dev)
            target = torch.rand(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(100 + view))
            out = render(cam, cloud, _Pipe(), torch.zeros(3, device=dev), cuda_type="pcheck_obb_sum")
            loss = l1_ssim_loss(out["render"], target, 0.2)
            if exchange:
                ex = multiview.OverlappedGradientExchange({"means3D": cloud._xyz, "opacities": cloud._opacity, "scales": cloud._scaling,
                                                           "rotations": cloud._rotation, "sh": cloud._features_dc, "sh_rest": cloud._features_rest}, ranges=4)
                with ex:
                    loss.backward()
                assert [c[0] for c in ex.calls] == [0, 1, 2, 3] and ex.calls[-1][2] == cloud._xyz.shape[0], ex.calls
                assert ex.exposed_ms() is not None
            else:
                loss.backward()
            torch.cuda.synchronize()
            return cloud
        summed = fused_view(rank, True)
        a_, b_ = fused_view(rank, False), fused_view(1 - rank, False)
        for i, (ps, pa, pb) in enumerate(zip(summed.parameters(), a_.parameters(), b_.parameters())):
            want = pa.grad + pb.grad
            err = float((ps.grad - want).abs().max()) / (float(want.abs().max()) + 1e-12)
            worst = max(worst, err)
            assert err < 2e-5 and float(want.abs().max()) > 0, ("overlapped", i, err)
        q.put((rank, int(vis.sum()), worst, None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001 -- the parent must see why a rank died
        import traceback
        q.put((rank, -1, -1.0, traceback.format_exc()))
        raise e


@pytest.mark.timeout(600)
def test_two_views_training_step_and_gradient_sum():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=500) for _ in range(world))
    for p in procs:
        p.join(60)
    for rank, nvis, worst, err in res:
        assert err is None, err
        assert nvis > 1000 and 0.0 <= worst < 2e-5
    assert all(p.exitcode == 0 for p in procs)


@pytest.mark.timeout(900)
def test_bench_train_mode_two_ranks_share_the_gpu():
    """`bench.py --gpus 2 --mode train` as the driver starts it (no torchrun environment): the launcher, two ranks, one
    result line from rank 0 with the exchange it made."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "train", "--steps", "2", "--warmup", "1",
                          "--points", "200000", "--width", "640", "--height", "352"], cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=800)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["unit"] == "ms/iter" and d["value"] > 0
    assert d["collective"]["mode"] in ("rows", "dense") and d["collective"]["bytes"] > 0 and d["collective_ms"] > 0
    assert d["config"]["parallelism"] == "views2" and d["config"]["gaussians"] == 200000
