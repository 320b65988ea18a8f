"""Developer tool (GPU): the training step (reference-shaped model, fused loss) with rasterizer.PREZERO_GRADIENTS off / on, interleaved,
on S-6M and S-6M-T. ms per step: wall clock of back-to-back steps; fwd / loss / bwd from events."""
import math, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fov3dgs_amd  # noqa
from fov3dgs_amd import rasterizer as rz, synthetic as syn
from fov3dgs_amd.gaussian_renderer import render
from fov3dgs_amd.loss_utils import l1_ssim_loss
dev = torch.device("cuda", 0)
cam = syn.camera_ring(0, 8).to(dev)
H, W = cam.image_height, cam.image_width
bg = torch.zeros(3, device=dev)


class Pipe:
    debug = False


for name, logit in (("S-6M", syn.OPACITY_LOGIT_S6M), ("S-6M-T", syn.OPACITY_LOGIT_S6MT)):
    cloud = syn.scene_bicycle_scale(opacity_logit=logit).to(dev).requires_grad_(True)
    model = syn.ReferenceShapedModel(cloud)
    target = torch.rand(3, H, W, device=dev)
    for rnd in range(3):
        for pre in (False, True):
            rz.PREZERO_GRADIENTS = pre
            n = 40
            evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(n + 5)]
            for it in range(n + 5):
                if it == 5:
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                for p in cloud.parameters():
                    p.grad = None
                e = evs[it]
                e[0].record()
                o = render(cam, model, Pipe(), bg, cuda_type="pcheck_obb_sum")
                e[1].record()
                loss = l1_ssim_loss(o["render"], target, 0.2)
                e[2].record()
                loss.backward()
                e[3].record()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / n * 1e3
            rows = np.array([(e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), e[2].elapsed_time(e[3])) for e in evs[5:]])
            print(f"{name} prezero={pre}: step {wall:.3f} ms; fwd / loss / bwd {np.median(rows, axis=0).round(3)}", flush=True)
    del cloud, model
    torch.cuda.empty_cache()
