#!/usr/bin/env python3
"""bench.py -- frames/sec of the foveated 1080p render on a synthetic bicycle-scale cloud.

Contract: `python bench.py --gpus N --steps K --warmup W` (for N > 1 launched by torch.distributed.run,
one rank per GPU). A step = one foveated frame through gaussian_renderer_fov.render() (4 layers,
alpha 0.05, Lissajous-moving gaze) of the seeded S-6M cloud at 1920x1080; with N ranks
each rank renders its own camera of an N-camera ring (weak scaling) and the frames are gathered on
rank 0 over RCCL, overlapped with the next frame. Rank 0 prints ONE JSON line.

Besides the headline value the line carries
  roofline      the dominant kernel's algorithmic bytes / its HIP-event duration vs the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (a scalar C port of the reference algorithm) on a bounded sample
  stages_ms     mean per-stage kernel time of the timed frames
  extra         non-foveated forward fps and training fwd+bwd ms on the same cloud (N = 1 only)
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import fov3dgs_amd  # noqa: E402,F401
from fov3dgs_amd import _native, multiview, synthetic as syn  # noqa: E402
from fov3dgs_amd.gaussian_renderer import render as render_plain  # noqa: E402
from fov3dgs_amd.gaussian_renderer_fov import render as render_fov  # noqa: E402
from fov3dgs_amd.profiling import StageTimer  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


class FrozenCloud:
    """Inference-time view of a cloud: activations evaluated once, getters return resident tensors."""

    def __init__(self, cloud):
        with torch.no_grad():
            self.get_xyz = cloud.get_xyz.detach()
            self.get_scaling = cloud.get_scaling.detach().contiguous()
            self.get_rotation = cloud.get_rotation.detach().contiguous()
            self.get_opacity = cloud.get_opacity.detach().contiguous()
            self.get_features = cloud.get_features.detach().contiguous()
            self.get_rest_features = cloud.get_rest_features.detach().contiguous()
        self.get_features_detach_rest = self.get_features
        self.active_sh_degree = cloud.active_sh_degree


class Pipe:
    debug = False


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def barrier_sync(world):
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()


def frame_stats(lib, vid, out_state, W, H, T):
    """V, D, D_single, D_blend of the last foveated forward call (reads the image workspace)."""
    num_rendered, radii, img = out_state
    rptr = lib.fr_image_ranges(vid, W, H, img.data_ptr())
    off = rptr - img.data_ptr()
    ranges = img[off:off + 8 * T].view(torch.int32).view(T, 2)
    lens = (ranges[:, 1] - ranges[:, 0]).to(torch.int64)
    lptr = lib.fr_image_tile_levels(W, H, img.data_ptr())
    off = lptr - img.data_ptr()
    lv = img[off:off + 20 * T].view(torch.float32).view(5, T)
    blend = lv[4] != 0
    d_blend = int(lens[blend].sum().item())
    d_all = int(lens.sum().item())
    return dict(V=int((radii > 0).sum().item()), D=d_all, D_single=d_all - d_blend, D_blend=d_blend,
                max_list=int(lens.max().item()), blend_tiles=int(blend.sum().item()))


def cpu_baseline(cloud_cpu, fov_cpu, cam, gaze, T_tiles, gx, gy):
    """Time the CPU oracle (scalar C port, 1 core) on a bounded sample: every Gaussian is preprocessed and
    culled (the per-frame part that does not depend on the window), binning + sort + blending are restricted
    to a 32x32-tile window around the image centre, and a 1x1 window isolates the window-independent part.
    Frame time = t(1x1) + (t(32x32) - t(1x1)) / 1023 * (tiles - 1); best of two runs each."""
    from oracle import oracle as orc
    from tests.helpers import cam_dict, scene_dict
    scene = scene_dict(cloud_cpu, "fov_pcheck_obb", fov_cpu)
    wx, wy = min(32, gx), min(32, gy)
    x0, y0 = (gx - wx) // 2, (gy - wy) // 2
    cd = cam_dict(cam, gaze=gaze, alpha=0.05)

    def timed(win):
        cd["tile_window"] = win
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            orc.forward("fov_pcheck_obb", scene, cd)
            best = min(best, time.perf_counter() - t0)
        return best
    t_total0 = time.perf_counter()
    t_win = timed((x0, y0, x0 + wx, y0 + wy))
    t_one = timed((x0 + wx // 2, y0 + wy // 2, x0 + wx // 2 + 1, y0 + wy // 2 + 1))
    per_tile = max(t_win - t_one, 0.0) / (wx * wy - 1)
    t_frame = t_one + per_tile * (T_tiles - 1)
    return dict(value=1.0 / t_frame, unit="frames/s", cores=1, kind="port",
                sample=(f"oracle/fovraster_oracle.c (scalar C, 1 thread): all {len(cloud_cpu)} Gaussians preprocessed+culled; "
                        f"{wx}x{wy}-tile centre window binned/sorted/blended in {t_win:.2f}s, 1x1 window {t_one:.2f}s; frame "
                        f"time extrapolated to {T_tiles} tiles = {t_frame:.2f}s"),
                seconds_measured=round(time.perf_counter() - t_total0, 2))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--points", type=int, default=6_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--gather", action="store_true",
                    help="N > 1: also collect every rank's frame on rank 0 (asynchronous RCCL gather overlapped with the next "
                         "frame). Off by default: the views are independent and the path has no exchange step.")
    args = ap.parse_args()

    rank, world, local_rank = multiview.init_distributed()
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    lib = _native.load()
    K, Wm = args.steps, args.warmup
    W, H = args.width, args.height
    gx, gy = (W + 15) // 16, (H + 15) // 16
    T = gx * gy

    t0 = time.time()
    cloud_cpu = syn.scene_bicycle_scale(P=args.points, seed=1)
    fov_cpu = syn.foveation_layers(cloud_cpu, seed=2)
    cloud = cloud_cpu.to(dev)
    pc = FrozenCloud(cloud)
    highest, shs_dcs, opac = [t.to(dev) for t in fov_cpu]
    n_views = max(world, 8)
    cam = syn.camera_ring(multiview.views_for_rank(rank, world, world)[0] * (n_views // world), n_views, W, H).to(dev)
    bg = torch.zeros(3, device=dev)
    if rank == 0:
        log(f"[bench] scene ready in {time.time() - t0:.1f}s: P={args.points} {W}x{H} world={world}")

    last = {}

    def step(i):
        gaze = syn.lissajous_gaze(i, 90)
        out = render_fov(cam, pc, bg, alpha=0.05, gazeArray=gaze, blending=True, highest_levels=highest,
                         shs_dcs=shs_dcs, opacities=opac, packed="auto")
        return out

    pending = None
    with torch.no_grad():
        for i in range(Wm):
            out = step(i)
            if world > 1 and args.gather:
                multiview.gather_images(out["render"], dst=0)
        barrier_sync(world)
        timer = StageTimer(K)
        t_start = time.perf_counter()
        with timer:
            for i in range(K):
                out = step(Wm + i)
                if world > 1 and args.gather:
                    if pending is not None:
                        pending[0].wait()
                    pending = multiview.gather_images(out["render"], dst=0, async_op=True) + (out["render"],)
            if pending is not None:
                pending[0].wait()
        barrier_sync(world)
        elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    stages = timer.stage_ms()
    timer.close()
    mean_ms = {k: float(np.mean([s[k] for s in stages])) for k in _native.STAGES}

    if rank != 0:
        return

    # ---- untimed post-pass: instance statistics of the same frames (for algorithmic bytes) ----
    vid = _native.VARIANT_FOV_PCHECK_OBB
    from fov3dgs_amd import rasterizer as rz
    stats = []
    with torch.no_grad():
        rs = None
        for i in range(0, K, max(1, K // 12)):
            gaze = syn.lissajous_gaze(Wm + i, 90)
            rs = rz.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg, 1.0,
                                                  cam.world_view_transform, cam.full_proj_transform, 3,
                                                  cam.camera_center, False, False)
            res = rz._forward_native(vid, rs, pc.get_xyz, pc.get_rest_features, torch.Tensor([]), opac, pc.get_scaling,
                                     pc.get_rotation, torch.Tensor([]), shs_dcs, highest, gaze, 0.05)
            torch.cuda.synchronize()
            stats.append(frame_stats(lib, vid, (res[0], res[2], res[5]), W, H, T))
        vm = cam.world_view_transform
        z = pc.get_xyz @ vm[:3, 2] + vm[3, 2]
        V_in = int((z > 0.2).sum().item())
    st = {k: float(np.mean([s[k] for s in stats])) for k in stats[0]}
    P, Px = args.points, W * H
    alg_bytes = {
        # SURVEY.md 8(d) per-unit figures x units of one launch
        # B_pre = 20 P + 224 V_in + 48 V (+24 V OBB axes), split over this build's two kernels: the cull pass streams
        # xyz/scale/rotation and writes radii; binning projects the survivors, reads opacity + SH and writes the
        # per-Gaussian record and the OBB axes
        "project": 20 * P + 28 * V_in,
        "bin": 196 * V_in + (48 + 24) * st["V"],
        "render": 32 * st["D_single"] + 52 * st["D_blend"] + 12 * Px,
        # this build's binning moves (depth,id) once per stage instead of a 6-pass radix sort
        "emit": 12 * st["D"] + 44 * st["V"],
        "tile_sort": 12 * st["D"],
        "tile_scan": 16 * T,
        "tile_levels": 20 * T,
    }
    dominant = max(("project", "bin", "render", "tile_sort", "emit"), key=lambda k: mean_ms[k])
    achieved = alg_bytes[dominant] / (mean_ms[dominant] * 1e-3) / 1e9
    traffic = None
    try:  # HBM bytes per launch of that kernel from the committed PMC pass (profiles/), if it exists
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc.json")))["kernels"]
        knames = {"render": ["k_render_fov"], "project": ["k_project"], "bin": ["k_bin", "k_hist_colscan"],
                  "tile_sort": ["k_tile_msort"], "emit": ["k_emit"]}[dominant]
        traffic = int(sum(2 * pmc[k]["FETCH_SIZE_KiB_per_frame"] + pmc[k]["WRITE_SIZE_KiB_per_frame"] for k in knames) * 1024)
    except Exception:
        pass
    roofline = dict(bound="hbm", kernel=dominant, achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic,
                    algorithmic_bytes=int(alg_bytes[dominant]), kernel_ms=round(mean_ms[dominant], 4),
                    per_kernel={k: dict(ms=round(mean_ms[k], 4), alg_GBs=round(alg_bytes[k] / max(mean_ms[k], 1e-9) / 1e6, 1))
                                for k in _native.STAGES})

    extra = {}
    if world == 1 and not args.no_extra:
        with torch.no_grad():
            for _ in range(3):
                render_plain(cam, pc, Pipe(), bg, cuda_type="pcheck_obb")
            torch.cuda.synchronize()
            n = 20
            t1 = time.perf_counter()
            for _ in range(n):
                render_plain(cam, pc, Pipe(), bg, cuda_type="pcheck_obb")
            torch.cuda.synchronize()
            extra["nonfov_forward_fps"] = round(n / (time.perf_counter() - t1), 2)
            # the same foveated frames without the packed copy of the static model that render() makes and caches by
            # itself (gaussian_renderer_fov._auto_packed; include/fovraster.h packed_geom / packed_colour)
            for i in range(Wm):
                render_fov(cam, pc, bg, alpha=0.05, gazeArray=syn.lissajous_gaze(i, 90), blending=True,
                           highest_levels=highest, shs_dcs=shs_dcs, opacities=opac, packed=None)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(K):
                render_fov(cam, pc, bg, alpha=0.05, gazeArray=syn.lissajous_gaze(Wm + i, 90), blending=True,
                           highest_levels=highest, shs_dcs=shs_dcs, opacities=opac, packed=None)
            torch.cuda.synchronize()
            extra["unpacked_model_fps"] = round(K / (time.perf_counter() - t1), 2)
        tr = cloud.requires_grad_(True)
        target = torch.rand(3, H, W, device=dev)
        ts = []
        for it in range(14):
            for p in tr.parameters():
                p.grad = None
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            o = render_plain(cam, tr, Pipe(), bg, cuda_type="pcheck_obb_sum")
            loss = (o["render"] - target).abs().mean()
            loss.backward()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t1) * 1e3)
        extra["train_fwd_bwd_ms"] = round(float(np.median(ts[2:])), 3)
        extra["train_loss"] = "L1 (rasterizer fwd+bwd incl. torch activations)"
        # the eff_finetune.py step with its real loss, 0.8 L1 + 0.2 (1 - SSIM): the fused HIP loss (csrc/loss.hip) next
        # to the reference's formulation (five grouped conv2d's + elementwise ops + autograd) in torch on the same GPU
        from fov3dgs_amd.loss_utils import l1_ssim_loss
        import torch.nn.functional as F
        g1 = torch.tensor([math.exp(-(i - 5) ** 2 / 4.5) for i in range(11)], device=dev)
        g1 = g1 / g1.sum()
        win = (g1[:, None] @ g1[None, :])[None, None].expand(3, 1, 11, 11).contiguous()

        def torch_loss(img, gt):
            a, b = img[None], gt[None]
            mu1, mu2 = F.conv2d(a, win, padding=5, groups=3), F.conv2d(b, win, padding=5, groups=3)
            s1 = F.conv2d(a * a, win, padding=5, groups=3) - mu1 * mu1
            s2 = F.conv2d(b * b, win, padding=5, groups=3) - mu2 * mu2
            s12 = F.conv2d(a * b, win, padding=5, groups=3) - mu1 * mu2
            m = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))
            return 0.8 * (img - gt).abs().mean() + 0.2 * (1.0 - m.mean())

        for name, fn in (("fused", lambda i, t: l1_ssim_loss(i, t, 0.2)), ("torch", torch_loss)):
            ts, tl = [], []
            for it in range(10):
                for p in tr.parameters():
                    p.grad = None
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                o = render_plain(cam, tr, Pipe(), bg, cuda_type="pcheck_obb_sum")
                fn(o["render"], target).backward()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t1) * 1e3)
                img = o["render"].detach().requires_grad_(True)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                fn(img, target).backward()
                torch.cuda.synchronize()
                tl.append((time.perf_counter() - t1) * 1e3)
            extra[f"train_step_l1_ssim_{name}_ms"] = round(float(np.median(ts[2:])), 3)
            extra[f"loss_fwd_bwd_{name}_ms"] = round(float(np.median(tl[2:])), 3)

    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(cloud_cpu, fov_cpu, cam, syn.lissajous_gaze(Wm, 90), T, gx, gy)
            cpu["value"] = round(cpu["value"], 5)
        except Exception as e:  # the baseline must never take the bench line down
            cpu = dict(value=None, unit="frames/s", cores=1, kind="port", sample=f"failed: {e}")

    fps = world * K / elapsed
    line = {
        "metric": "frames/sec at 1080p foveated (bicycle-scale)", "value": round(fps, 3), "unit": "frames/s",
        "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": round(elapsed / K * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "S-6M bicycle-scale cloud, 4-layer foveated render (fov_pcheck_obb), moving gaze, "
                               "one camera per GPU" + (", frames gathered on rank 0" if (world > 1 and args.gather) else ""),
                   "gaussians": P, "width": W, "height": H, "alpha": 0.05, "sh_degree": 3,
                   "visible": int(st["V"]), "in_front": V_in, "instances": int(st["D"]),
                   "instances_blend_tiles": int(st["D_blend"]), "max_tile_list": int(st["max_list"]),
                   "model_layout": "static model: render() packs it once during warm-up (packed_geom/packed_colour, "
                                   "bit-identical image); extra.unpacked_model_fps = same frames without",
                   "parallelism": f"views{world}"},
        "roofline": roofline,
        "cpu_baseline": cpu,
        "stages_ms": {k: round(v, 4) for k, v in mean_ms.items()},
        "extra": extra,
    }
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
