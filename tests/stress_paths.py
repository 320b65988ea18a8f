"""Randomised self-consistency sweep of the autograd-level input forms (not collected by pytest): for random small scenes the training
rasterizer is called through every form this package offers -- SH coefficients whole / as the two tensors a model stores, activated /
raw parameters, dense / row-sparse gradients, with / without the per-Gaussian statistics, the packed static-model layout (forward) --
and every form must give the image (bit for bit where the kernels are the same) and the gradients of the plain call.
usage: python tests/stress_paths.py [rounds=40] [seed=0]"""
import math, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import small_cloud, small_camera, syn
from tests.checks import grad_stats
from fov3dgs_amd import rasterizer as rz
from fov3dgs_amd.activations import activate
from fov3dgs_amd.diff_gaussian_rasterization_pcheck_obb_sum import GaussianRasterizationSettings, GaussianRasterizer

rz.POISON_GRADIENTS = True  # every gradient tensor starts as NaN: an element the library fails to write shows in the comparisons
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = "cuda:0"
names = ("xyz", "scaling", "rotation", "opacity", "features_dc", "features_rest")
bad = 0
for r in range(rounds):
    P = int(rng.choice([1, 2, 63, 64, 65, 129, 700, 4000, 12000]))
    W, H = int(rng.integers(17, 700)), int(rng.integers(17, 500))
    seed = int(rng.integers(1 << 30))
    deg = int(rng.integers(0, 4))
    cam = small_camera(W, H).to(dev)
    rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.tensor([0.1, 0.2, 0.3], device=dev),
                                       float(rng.choice([1.0, 0.5, 2.0])), cam.world_view_transform, cam.full_proj_transform, deg, cam.camera_center, False, False)
    if int(os.environ.get("STRESS_ONLY", "-1")) not in (-1, r):  # replay one round of a sweep (the random numbers above are drawn for all)
        continue
    w = torch.randn(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(seed & 0xffff))

    def run(form):
        cloud = (small_cloud(P, seed) if P >= 8 else syn.scene_1k(P=P, seed=seed)).to(dev).requires_grad_(True)
        kw = {}
        if form in ("raw", "raw_sparse"):
            s, q, o = cloud._scaling, cloud._rotation, cloud._opacity
            kw["raw_activations"] = True
        else:
            s, q, o = activate(cloud._scaling, cloud._rotation, cloud._opacity)
        shs = torch.cat((cloud._features_dc, cloud._features_rest), dim=1) if form == "cat" else (cloud._features_dc, cloud._features_rest)
        if form == "raw_sparse":
            kw["row_sparse"] = True
        if form == "nostats":
            kw["want_stats"] = False
        if form == "packed":
            with torch.no_grad():
                kw["packed"] = rz.pack_model(cloud._xyz, s.detach(), q.detach(), o.detach(), shs=cloud._features_dc, shs_rest=cloud._features_rest)
        m2d = torch.zeros_like(cloud._xyz, requires_grad=True)
        out = GaussianRasterizer(rs)(means3D=cloud._xyz, means2D=m2d, opacities=o, shs=shs, scales=s, rotations=q, **kw)
        grads = None
        if form != "packed":
            (out[0] * w).sum().backward()
            grads = {n: (getattr(cloud, "_" + n).grad.to_dense() if getattr(cloud, "_" + n).grad.is_sparse else getattr(cloud, "_" + n).grad).clone() for n in names}
            grads["means2D"] = m2d.grad.to_dense().clone() if m2d.grad.is_sparse else m2d.grad.clone()
        torch.cuda.synchronize()
        return out, grads
    base, gb = run("split")
    ok, notes = True, []
    # for context: the SAME form once more. Two backward passes differ by the order their float atomics retire in, and the near-camera
    # splats of some scenes turn that into 1e-4 of a tensor's norm (scales, positions, rotations: the covariance chain) -- between two
    # FORMS, whose other kernels shift the waves' timing, rather than between two runs of one form (1e-5: the order then nearly repeats;
    # seeds 71 / 72, rounds 66, 342, 354: which forms land 1.2-1.8e-4 away changes from run to run). Hence 5e-4 on the norm: a wrong
    # row is 1e-2 and more.
    _, g2 = run("split")
    noise = {n: grad_stats(g2[n].cpu().numpy(), gb[n].cpu().numpy(), rtol=1e-4) for n in gb}
    for form in ("cat", "raw", "raw_sparse", "nostats", "packed"):
        o, g = run(form)
        same = torch.equal(o[0], base[0]) and torch.equal(o[1], base[1])
        if not same:
            d = (o[0] - base[0]).abs().max().item()
            notes.append(f"{form}: image differs by {d:.2e}")
            ok = ok and d < 1e-6 and torch.equal(o[1], base[1])  # (the packed / raw forms are documented bit-identical: anything else is noted)
        if form not in ("nostats", "packed") and len(o) == 4:
            ok = ok and torch.equal(o[2], base[2])
        if g is not None:
            for n in g:
                st = grad_stats(g[n].cpu().numpy(), gb[n].cpu().numpy(), rtol=1e-4)
                if not (st["frac_bad"] <= max(5e-3, 10.0 / max(st["rows_with_gradient"], 1), 3.0 * noise[n]["frac_bad"]) and
                        st["rel_l2"] <= max(5e-4, 3.0 * noise[n]["rel_l2"]) and np.isfinite(g[n].cpu().numpy()).all()):
                    ok = False
                    notes.append(f"{form} {n}: rows outside 1e-4 {st['frac_bad']:.1e}, rel L2 {st['rel_l2']:.1e} (the same form twice: {noise[n]['frac_bad']:.1e}, {noise[n]['rel_l2']:.1e})")
    print(f"{r:3d} P={P:6d} {W}x{H} deg={deg} visible {int((base[1] > 0).sum()):6d} -> {'ok' if ok else 'MISMATCH'} {'; '.join(notes)}", flush=True)
    bad += 0 if ok else 1
print("mismatches:", bad)
sys.exit(1 if bad else 0)
