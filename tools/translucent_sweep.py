"""Developer tool (GPU): pick the opacity distribution of the S-6M-T workload. For every candidate (mean, std) of the opacity
logit: fraction of the frame's list entries the blend fetches (fr_forward_args.list_consumed) on the plain training frame and on
the nine foveated bench gazes, stage times of the foveated frames and the time of a raw-parameter training step."""
import math, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fov3dgs_amd  # noqa
from fov3dgs_amd import _native, rasterizer as rz, synthetic as syn
from fov3dgs_amd.gaussian_renderer import render as render_plain
from fov3dgs_amd.loss_utils import l1_ssim_loss
from fov3dgs_amd.profiling import StageTimer

dev = torch.device("cuda", 0)
GAZES = [(0.25 * i, 0.25 * j) for i in range(1, 4) for j in range(1, 4)]
cands = [tuple(float(x) for x in a.split(",")) for a in sys.argv[1:]] or [(1.0, 2.0), (-1.5, 1.5), (-2.5, 1.5), (-3.5, 1.5)]
cam = syn.camera_ring(0, 8).to(dev)
W, H = cam.image_width, cam.image_height
T = ((W + 15) // 16) * ((H + 15) // 16)
bg = torch.zeros(3, device=dev)
rs = rz.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg, 1.0, cam.world_view_transform,
                                      cam.full_proj_transform, 3, cam.camera_center, False, False)
E = torch.Tensor([])


class Pipe:
    debug = False


for mean, std in cands:
    cpu = syn.scene_bicycle_scale(opacity_logit=(mean, std))
    fov = [t.to(dev) for t in syn.foveation_layers(cpu, seed=2)]
    cloud = cpu.to(dev)
    with torch.no_grad():
        xyz, sc, rot, op = cloud.get_xyz, cloud.get_scaling.contiguous(), cloud.get_rotation.contiguous(), cloud.get_opacity.contiguous()
        rest, sh = cloud._features_rest.contiguous(), cloud.get_features.contiguous()
        cons = torch.zeros(T, dtype=torch.int32, device=dev)
        r = rz._forward_native(_native.VARIANT_PCHECK_OBB_SUM, rs, xyz, sh, E, op, sc, rot, E, persistent=True, list_consumed=cons)
        torch.cuda.synchronize()
        print(f"logit N({mean},{std}^2): plain D={r[0]} consumed {cons.sum().item() / r[0]:.3f} longest {r[0] and int(cons.max())}", flush=True)
        fr = []
        for g in GAZES:
            r = rz._forward_native(_native.VARIANT_FOV_PCHECK_OBB, rs, xyz, rest, E, fov[2], sc, rot, E, fov[1], fov[0], g, 0.05, persistent=True, list_consumed=cons)
            torch.cuda.synchronize()
            fr.append(cons.sum().item() / r[0])
        print(f"   foveated consumed per gaze {[round(x, 3) for x in fr]} mean {np.mean(fr):.3f}", flush=True)
        for _ in range(9):
            rz._forward_native(_native.VARIANT_FOV_PCHECK_OBB, rs, xyz, rest, E, fov[2], sc, rot, E, fov[1], fov[0], GAZES[_], 0.05, persistent=True)
        timer = StageTimer(27)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with timer:
            for i in range(27):
                rz._forward_native(_native.VARIANT_FOV_PCHECK_OBB, rs, xyz, rest, E, fov[2], sc, rot, E, fov[1], fov[0], GAZES[i % 9], 0.05, persistent=True)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 27 * 1e3
        st = timer.stage_ms(); timer.close()
        print(f"   foveated frame {wall:.3f} ms; stages " + " ".join(f"{k}={np.mean([s[k] for s in st]):.3f}" for k in _native.STAGES), flush=True)
    tr = cloud.requires_grad_(True)
    tr.fuse_activations = True
    target = torch.rand(3, H, W, device=dev)
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(25)]
    for it in range(25):
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
    rows = np.array([(e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), e[2].elapsed_time(e[3])) for e in evs[5:]])
    print(f"   training step fwd/loss/bwd ms {np.median(rows, axis=0).round(3)}; rows with a gradient {(tr._opacity.grad != 0).sum().item()}", flush=True)
    del cloud, tr, cpu, fov
    torch.cuda.empty_cache()
