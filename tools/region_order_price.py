"""Developer tool (GPU): what would spatially coherent item order buy? The S-6M model is permuted so that Gaussians whose projected
centres (bench camera) fall into the same coarse screen region are consecutive (stable: increasing index inside a region; the
off-screen ones last) -- i.e. the library's index-ordered item list IS region-ordered, at no cost. Stage times of the foveated bench
frames with the model as it is / region-ordered with regions of R x R tiles. Pricing only (the order is camera-specific)."""
import math, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fov3dgs_amd  # noqa
from fov3dgs_amd import _native, rasterizer as rz, synthetic as syn
from fov3dgs_amd.profiling import StageTimer

dev = torch.device("cuda", 0)
GAZES = [(0.25 * i, 0.25 * j) for i in range(1, 4) for j in range(1, 4)]
cam = syn.camera_ring(0, 8).to(dev)
W, H = cam.image_width, cam.image_height
bg = torch.zeros(3, device=dev)
rs = rz.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg, 1.0, cam.world_view_transform,
                                      cam.full_proj_transform, 3, cam.camera_center, False, False)
E = torch.Tensor([])
cpu = syn.scene_bicycle_scale()
fov_cpu = syn.foveation_layers(cpu, seed=2)


def run(tag, perm):
    cloud = cpu.to(dev)
    fov = [t.to(dev) for t in fov_cpu]
    with torch.no_grad():
        xyz, sc, rot = cloud.get_xyz, cloud.get_scaling.contiguous(), cloud.get_rotation.contiguous()
        rest = cloud._features_rest.contiguous()
        if perm is not None:
            xyz, sc, rot, rest = xyz[perm].contiguous(), sc[perm].contiguous(), rot[perm].contiguous(), rest[perm].contiguous()
            fov = [t[perm].contiguous() for t in fov]
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
                r = f(GAZES[i % 9])
        torch.cuda.synchronize()
        st = timer.stage_ms(); timer.close()
    print(f"{tag}: {np.median(walls):.4f} ms D={r[0]} image sum {float(r[1].double().sum()):.6f} [" + " ".join(f"{k[:5]}={np.mean([s[k] for s in st]):.3f}" for k in _native.STAGES) + "]", flush=True)


run("model as it is", None)
with torch.no_grad():
    x = cpu.get_xyz.to(dev)
    pm = cam.full_proj_transform
    hom = torch.cat([x, torch.ones_like(x[:, :1])], 1) @ pm
    w = 1.0 / (hom[:, 3] + 1e-7)
    px = ((hom[:, 0] * w + 1) * W - 1) * 0.5
    py = ((hom[:, 1] * w + 1) * H - 1) * 0.5
    vm = cam.world_view_transform
    z = x @ vm[:3, 2] + vm[3, 2]
    for R in (8, 4, 2, 1):
        gx, gy = (W + 16 * R - 1) // (16 * R), (H + 16 * R - 1) // (16 * R)
        rx = torch.clamp((px / (16 * R)).floor().long(), 0, gx - 1)
        ry = torch.clamp((py / (16 * R)).floor().long(), 0, gy - 1)
        reg = ry * gx + rx
        off = (z <= 0.2) | (px < -200) | (px > W + 200) | (py < -200) | (py > H + 200)
        reg = torch.where(off, torch.full_like(reg, gx * gy), reg)
        perm = torch.sort(reg, stable=True).indices
        run(f"regions of {R}x{R} tiles ({gx * gy} regions)", perm)
