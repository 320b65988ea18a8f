"""Developer tool (GPU): per-kernel times of the training step (forward stages, k_render_bwd, k_preprocess_bwd, k_fill_zero) on S-6M
or, with `T` as first argument, S-6M-T; and of the foveated frame's stages. usage: python tools/train_stages.py [T] [nostats]"""
import math, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fov3dgs_amd  # noqa
from fov3dgs_amd import _native, rasterizer as rz, synthetic as syn
from fov3dgs_amd.gaussian_renderer import render as render_plain
from fov3dgs_amd.loss_utils import l1_ssim_loss
from fov3dgs_amd.profiling import StageTimer, BackwardTimer

dev = torch.device("cuda", 0)
GAZES = [(0.25 * i, 0.25 * j) for i in range(1, 4) for j in range(1, 4)]
logit = syn.OPACITY_LOGIT_S6MT if "T" in sys.argv[1:] else syn.OPACITY_LOGIT_S6M
kw = {"want_stats": False} if "nostats" in sys.argv[1:] else {}
cam = syn.camera_ring(0, 8).to(dev)
W, H = cam.image_width, cam.image_height
bg = torch.zeros(3, device=dev)
rs = rz.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg, 1.0, cam.world_view_transform,
                                      cam.full_proj_transform, 3, cam.camera_center, False, False)
E = torch.Tensor([])


class Pipe:
    debug = False


cpu = syn.scene_bicycle_scale(opacity_logit=logit)
fov = [t.to(dev) for t in syn.foveation_layers(cpu, seed=2)]
cloud = cpu.to(dev)
with torch.no_grad():
    xyz, sc, rot = cloud.get_xyz, cloud.get_scaling.contiguous(), cloud.get_rotation.contiguous()
    rest = cloud._features_rest.contiguous()
    for i in range(9):
        rz._forward_native(_native.VARIANT_FOV_PCHECK_OBB, rs, xyz, rest, E, fov[2], sc, rot, E, fov[1], fov[0], GAZES[i], 0.05, persistent=True)
    for rep in range(2):
        timer = StageTimer(27)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with timer:
            for i in range(27):
                rz._forward_native(_native.VARIANT_FOV_PCHECK_OBB, rs, xyz, rest, E, fov[2], sc, rot, E, fov[1], fov[0], GAZES[i % 9], 0.05, persistent=True)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 27 * 1e3
        st = timer.stage_ms(); timer.close()
        print(f"logit {logit}: foveated frame {wall:.3f} ms; " + " ".join(f"{k}={np.mean([s[k] for s in st]):.3f}" for k in _native.STAGES), flush=True)
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
        o = render_plain(cam, tr, Pipe(), bg, cuda_type="pcheck_obb_sum", **kw)
        e[1].record()
        loss = l1_ssim_loss(o["render"], target, 0.2)
        e[2].record()
        loss.backward()
        e[3].record()
torch.cuda.synchronize()
rows = np.array([(e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), e[2].elapsed_time(e[3])) for e in evs[4:]])
print(f"training step fwd/loss/bwd ms {np.median(rows, axis=0).round(3)} {kw}")
print("  forward stages: " + " ".join(f"{k}={np.median([r[k] for r in ft.stage_ms()[4:]]):.3f}" for k in _native.STAGES))
print("  backward: " + " ".join(f"{k}={np.median([r[k] for r in bt.stage_ms()[4:]]):.3f}" for k in ("render_bwd", "preprocess_bwd", "fill_zero")))
