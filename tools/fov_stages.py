"""Developer tool (GPU): ONE line -- the foveated bench frames' wall time and stage times on S-6M and on S-6M-T (for tools/ab_run.sh).
usage: python tools/fov_stages.py [train]   (train: also the training step's kernels on both clouds)"""
import math, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fov3dgs_amd  # noqa
from fov3dgs_amd import _native, rasterizer as rz, synthetic as syn
from fov3dgs_amd.profiling import StageTimer, BackwardTimer

dev = torch.device("cuda", 0)
GAZES = [(0.25 * i, 0.25 * j) for i in range(1, 4) for j in range(1, 4)]
cam = syn.camera_ring(0, 8).to(dev)
W, H = cam.image_width, cam.image_height
bg = torch.zeros(3, device=dev)
rs = rz.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg, 1.0, cam.world_view_transform,
                                      cam.full_proj_transform, 3, cam.camera_center, False, False)
E = torch.Tensor([])
out = []
train = "train" in sys.argv[1:]


class Pipe:
    debug = False


for name, logit in (("S-6M", syn.OPACITY_LOGIT_S6M), ("S-6M-T", syn.OPACITY_LOGIT_S6MT)):
    cpu = syn.scene_bicycle_scale(opacity_logit=logit)
    fov = [t.to(dev) for t in syn.foveation_layers(cpu, seed=2)]
    cloud = cpu.to(dev)
    with torch.no_grad():
        xyz, sc, rot = cloud.get_xyz, cloud.get_scaling.contiguous(), cloud.get_rotation.contiguous()
        rest = cloud._features_rest.contiguous()
        f = lambda g: rz._forward_native(_native.VARIANT_FOV_PCHECK_OBB, rs, xyz, rest, E, fov[2], sc, rot, E, fov[1], fov[0], g, 0.05, persistent=True)
        for i in range(18):
            f(GAZES[i % 9])
        walls = []
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(63):
                f(GAZES[i % 9])
            torch.cuda.synchronize()
            walls.append((time.perf_counter() - t0) / 63 * 1e3)
        timer = StageTimer(27)
        with timer:
            for i in range(27):
                f(GAZES[i % 9])
        torch.cuda.synchronize()
        st = timer.stage_ms(); timer.close()
    out.append(f"{name}: {np.median(walls):.4f} ms [" + " ".join(f"{k[:4]}={np.mean([s[k] for s in st]):.3f}" for k in _native.STAGES) + "]")
    if train:
        from fov3dgs_amd.gaussian_renderer import render as render_plain
        from fov3dgs_amd.loss_utils import l1_ssim_loss
        tr = cloud.requires_grad_(True)
        tr.fuse_activations = True
        target = torch.rand(3, H, W, device=dev)
        n = 14
        ft, bt = StageTimer(n), BackwardTimer(n)
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(n)]
        with ft, bt:
            for it in range(n):
                for p in tr.parameters():
                    p.grad = None
                e = evs[it]
                e[0].record()
                o = render_plain(cam, tr, Pipe(), bg, cuda_type="pcheck_obb_sum")
                e[1].record()
                loss = l1_ssim_loss(o["render"], target, 0.2)
                e[2].record()
                loss.backward()
                e[3].record()
        torch.cuda.synchronize()
        rows = np.array([(e[0].elapsed_time(e[1]), e[2].elapsed_time(e[3])) for e in evs[4:]])
        med = np.median(rows, axis=0)
        out.append(f"train fwd {med[0]:.3f} bwd {med[1]:.3f} [" + " ".join(f"{k[:4]}={np.median([r[k] for r in ft.stage_ms()[4:]]):.3f}" for k in _native.STAGES) + " | "
                   + " ".join(f"{k}={np.median([r[k] for r in bt.stage_ms()[4:]]):.3f}" for k in ("render_bwd", "preprocess_bwd", "fill_zero")) + "]")
        del tr
    del cloud, cpu, fov
    torch.cuda.empty_cache()
print(" || ".join(out))
