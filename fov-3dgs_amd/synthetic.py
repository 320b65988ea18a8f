"""Seeded synthetic Gaussian clouds and cameras: the measurement workload.

There is no dataset or checkpoint on the GPU box, so tests and bench.py render
synthetic clouds of the same shape as the reference's inputs. ``GaussianCloud``
exposes the getters the reference's render() reads from its GaussianModel
(fov3dgs/scene/gaussian_model.py:200-240): raw parameters plus the exp / sigmoid /
normalize activations and the dc|rest SH concatenation.
"""
import math

import numpy as np
import torch

from .cameras import MiniCam, look_at

# bicycle per-level Gaussian counts (reference fov3dgs/pnum/ours-Q/bicycle.txt:1-4)
# => fraction of Gaussians whose highest level is 0/1/2/3
LEVEL_FRACTIONS = (0.599, 0.183, 0.043, 0.175)


class GaussianCloud:
    def __init__(self, xyz, features_dc, features_rest, scaling, rotation, opacity, sh_degree=3):
        self._xyz, self._features_dc, self._features_rest = xyz, features_dc, features_rest
        self._scaling, self._rotation, self._opacity = scaling, rotation, opacity
        self.active_sh_degree = sh_degree
        self.max_sh_degree = sh_degree

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    @property
    def get_activated(self):
        """(scaling, rotation, opacity) in one fused pass each way on the GPU (activations.py; extension, picked up by
        gaussian_renderer.render()); the three getters above on the CPU."""
        if self._scaling.is_cuda:
            from .activations import activate
            return activate(self._scaling, self._rotation, self._opacity)
        return self.get_scaling, self.get_rotation, self.get_opacity

    # opt-in (set `fuse_activations = True` on the cloud): hand the raw parameters to the rasterizer, which applies the
    # activations in its kernels (rasterizer.GaussianRasterizer(..., raw_activations=True)); GPU only
    fuse_activations = False

    @property
    def get_raw_activation_params(self):
        if self.fuse_activations and self._scaling.is_cuda:
            return self._scaling, self._rotation, self._opacity
        return None

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_rest_features(self):
        return self._features_rest

    @property
    def get_features_split(self):
        """(features_dc [P,1,3], features_rest [P,M-1,3]) as stored: the rasterizer takes them as they are
        (fr_forward_args.shs_rest), which saves get_features' torch.cat and its backward every step."""
        return self._features_dc, self._features_rest

    @property
    def get_features_split_detach_rest(self):
        return self._features_dc, self._features_rest.detach()

    @property
    def get_features_detach_rest(self):
        return torch.cat((self._features_dc, self._features_rest.detach()), dim=1)

    def parameters(self):
        return [self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation, self._opacity]

    def to(self, device):
        return GaussianCloud(*[p.to(device) for p in self.parameters()], sh_degree=self.active_sh_degree)

    def requires_grad_(self, flag=True):
        for p in self.parameters():
            p.requires_grad_(flag)
        return self

    def __len__(self):
        return self._xyz.shape[0]


class ReferenceGetterModel:
    """A model that offers ONLY what the reference's GaussianModel offers (fov3dgs/scene/gaussian_model.py:200-240): get_xyz,
    get_scaling = exp, get_rotation = normalize, get_opacity = sigmoid, get_features = torch.cat((dc, rest), dim=1) and the
    detach_rest form masking uses -- none of this package's extension getters (get_activated, get_raw_activation_params,
    get_features_split). What a maintainer gets from switching the imports and nothing else; shares the
    parameter tensors of `cloud`."""

    def __init__(self, cloud):
        self._c = cloud
        self.active_sh_degree = cloud.active_sh_degree
        self.max_sh_degree = cloud.max_sh_degree

    @property
    def get_xyz(self):
        return self._c._xyz

    @property
    def get_scaling(self):
        return torch.exp(self._c._scaling)

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._c._rotation)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._c._opacity)

    @property
    def get_features(self):
        return torch.cat((self._c._features_dc, self._c._features_rest), dim=1)

    @property
    def get_features_detach_rest(self):
        return torch.cat((self._c._features_dc, self._c._features_rest.detach()), dim=1)

    @property
    def get_rest_features(self):
        return self._c._features_rest

    def parameters(self):
        return self._c.parameters()


class ReferenceShapedModel(ReferenceGetterModel):
    """A model with the ATTRIBUTES of the reference's GaussianModel as well as its getters (scene/gaussian_model.py:33-50: the raw
    parameter tensors _xyz / _features_dc / _features_rest / _scaling / _rotation / _opacity and the activation functions
    scaling_activation / opacity_activation / rotation_activation), which gaussian_renderer.render() recognises
    (FAST_REFERENCE_MODEL): what a maintainer's own GaussianModel object looks like to render()."""

    def __init__(self, cloud):
        super().__init__(cloud)
        self._xyz, self._features_dc, self._features_rest = cloud._xyz, cloud._features_dc, cloud._features_rest
        self._scaling, self._rotation, self._opacity = cloud._scaling, cloud._rotation, cloud._opacity
        self.scaling_activation, self.opacity_activation = torch.exp, torch.sigmoid
        self.rotation_activation = torch.nn.functional.normalize

    # the getters as the reference's class has them (scene/gaussian_model.py:200-240): one-line expressions of the attributes above;
    # render() checks that this is what they are before it bypasses them (gaussian_renderer._getter_fingerprint_ok)
    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_scaling(self):
        return self.scaling_activation(self._scaling)

    @property
    def get_rotation(self):
        return self.rotation_activation(self._rotation)

    @property
    def get_opacity(self):
        return self.opacity_activation(self._opacity)

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_features_detach_rest(self):
        return torch.cat((self._features_dc, self._features_rest.detach()), dim=1)


class MaskedOpacityModel(ReferenceShapedModel):
    """A subclass that keeps every attribute of the reference's model but overrides ONE getter (opacity times a learned mask, as the
    reference's masking experiments do with get_mask, scene/gaussian_model.py:212-214): render() must not bypass its getters."""

    def __init__(self, cloud, mask_logit):
        super().__init__(cloud)
        self._mask = mask_logit

    @property
    def get_opacity(self):
        return self.opacity_activation(self._opacity) * torch.sigmoid(self._mask)


def _gen(seed):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return g


def scene_1k(P=1000, seed=0, device="cpu"):
    """S-1k: P random Gaussians in front of an identity camera (BASELINE config 1)."""
    g = _gen(seed)
    xyz = torch.rand(P, 3, generator=g) * 2 - 1
    xyz[:, 2] += 4.0
    scaling = math.log(0.05) + 0.5 * torch.randn(P, 3, generator=g)
    rotation = torch.randn(P, 4, generator=g)
    opacity = 1.5 * torch.randn(P, 1, generator=g)
    f_dc = torch.randn(P, 1, 3, generator=g)
    f_rest = 0.1 * torch.randn(P, 15, 3, generator=g)
    return GaussianCloud(xyz, f_dc, f_rest, scaling, rotation, opacity).to(device)


def camera_1k(width=256, height=256, fov_deg=60.0, device="cpu"):
    fov = math.radians(fov_deg)
    return MiniCam(np.eye(3), np.zeros(3), fov, fov, width, height, device=device)


OPACITY_LOGIT_S6M = (1.0, 2.0)     # opacity = sigmoid(N(1, 2^2)) -- median alpha 0.73: pixels saturate within 9-20 % of their lists
OPACITY_LOGIT_S6MT = (-3.5, 1.0)   # S-6M-T ("translucent"): same geometry and seeds, median alpha 0.029 (see scene_translucent)


def scene_bicycle_scale(P=6_000_000, seed=1, device="cpu", scale_log_mean=math.log(0.01), opacity_logit=OPACITY_LOGIT_S6M):
    """S-6M ("bicycle-scale"): ground annulus + dome shell + a dense centre, y is down-positive.
    opacity_logit = (mean, std) of the normal the opacity logits are drawn from (the draws themselves are the same for any
    choice: two clouds that differ only here have identical geometry, SH and per-Gaussian random numbers)."""
    g = _gen(seed)
    n_ground = int(0.70 * P)
    n_dome = int(0.25 * P)
    n_centre = P - n_ground - n_dome
    # ground-plane annulus r in [1,8], |y| < 0.3 (uniform in area)
    r = torch.sqrt(torch.rand(n_ground, generator=g) * (64 - 1) + 1)
    th = torch.rand(n_ground, generator=g) * (2 * math.pi)
    ground = torch.stack([r * torch.cos(th), (torch.rand(n_ground, generator=g) * 2 - 1) * 0.3, r * torch.sin(th)], 1)
    # upper dome shell r in [6,12] (y <= 0 is "up")
    d = torch.randn(n_dome, 3, generator=g)
    d = d / d.norm(dim=1, keepdim=True)
    d[:, 1] = -d[:, 1].abs()
    rr = 6 + 6 * torch.rand(n_dome, 1, generator=g)
    dome = d * rr
    # dense centre r < 1
    c = torch.randn(n_centre, 3, generator=g)
    c = c / c.norm(dim=1, keepdim=True) * torch.rand(n_centre, 1, generator=g).pow(1 / 3)
    xyz = torch.cat([ground, dome, c], 0)
    xyz = xyz[torch.randperm(P, generator=g)]
    scaling = scale_log_mean + 0.7 * torch.randn(P, 1, generator=g) + 0.5 * torch.randn(P, 3, generator=g)
    rotation = torch.randn(P, 4, generator=g)
    opacity = opacity_logit[0] + opacity_logit[1] * torch.randn(P, 1, generator=g)
    f_dc = torch.randn(P, 1, 3, generator=g)
    f_rest = 0.1 * torch.randn(P, 15, 3, generator=g)
    return GaussianCloud(xyz.contiguous(), f_dc, f_rest, scaling, rotation, opacity).to(device)


def scene_translucent(P=6_000_000, seed=1, device="cpu", opacity_logit=OPACITY_LOGIT_S6MT):
    """S-6M-T: the S-6M cloud with opacities drawn so that the blend CONSUMES its lists -- on S-6M (median alpha 0.73) every pixel
    saturates after 9 % (plain) to 20 % (foveated) of its tile's list, which no trained model does: the reference's loops run to
    the end of most lists (RS forward.cu:349-421, R0 backward.cu:455-557). Same positions, scales, rotations, SH and seeds.
    Opacity logit ~ N(-3.5, 1) (tools/translucent_sweep.py, measured on the bench camera): the blend fetches 0.72 of the training
    frame's 16.9 M instances (S-6M: 0.10) and 0.75-1.0 of the nine foveated bench frames' (S-6M: 0.16-0.25), walks lists 5 248
    entries = 20 rounds of 256 deep, and 1.03 M Gaussians receive a gradient (S-6M: 131 k). Candidates: N(-2.5, 1.5): 0.29 / 0.61,
    341 k rows; N(-3.5, 1.5): 0.54 / 0.81, 626 k; N(-4, 1): 0.93 / 0.98, 1.49 M; N(-4.5, 1): everything consumed."""
    return scene_bicycle_scale(P=P, seed=seed, device=device, opacity_logit=opacity_logit)


def camera_ring(index=0, n=8, width=1920, height=1080, fovx_deg=62.0, radius=4.0, height_above=1.0, device="cpu"):
    """Camera `index` of `n` on a circle of `radius`, `height_above` the ground, looking at the origin."""
    th = 2 * math.pi * index / n
    eye = (radius * math.cos(th), -height_above, radius * math.sin(th))
    R, t = look_at(eye, (0.0, 0.0, 0.0))
    fovx = math.radians(fovx_deg)
    fx = width / (2 * math.tan(fovx / 2))
    fovy = 2 * math.atan(height / (2 * fx))
    return MiniCam(R, t, fovx, fovy, width, height, device=device)


def foveation_layers(cloud, seed=2, fractions=LEVEL_FRACTIONS, device=None):
    """Per-Gaussian foveation inputs in the format fov3dgs/compose_models.py:51-80 writes:
    highest_levels f32[P,1], shs_dcs f32[P,4,3], opacities f32[P,4]. Level i carries level
    i-1's value forward and perturbs it for the Gaussians that still exist at level i."""
    g = _gen(seed)
    P = len(cloud)
    dev = cloud.get_xyz.device if device is None else device
    u = torch.rand(P, generator=g)
    edges = torch.tensor(fractions).cumsum(0)
    highest = torch.bucketize(u, edges[:-1]).to(torch.float32).unsqueeze(1)
    dc0 = cloud._features_dc.detach().cpu()[:, 0, :]
    op0 = torch.sigmoid(cloud._opacity.detach().cpu())[:, 0]
    L = len(fractions)
    shs_dcs = torch.zeros(P, L, 3)
    opac = torch.ones(P, L)
    shs_dcs[:, 0] = dc0
    opac[:, 0] = op0
    for i in range(1, L):
        alive = (highest[:, 0] >= i)
        shs_dcs[:, i] = shs_dcs[:, i - 1]
        opac[:, i] = opac[:, i - 1]
        shs_dcs[alive, i] += 0.05 * torch.randn(int(alive.sum()), 3, generator=g)
        opac[alive, i] = (opac[alive, i] + 0.05 * torch.randn(int(alive.sum()), generator=g)).clamp(0.0, 1.0)
    return highest.to(dev), shs_dcs.to(dev), opac.to(dev)


def lissajous_gaze(frame, n_frames=90):
    t = frame / n_frames
    return (0.5 + 0.25 * math.sin(2 * math.pi * t), 0.5 + 0.25 * math.sin(3 * math.pi * t))
