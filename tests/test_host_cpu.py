"""CPU-side checks: the C-ABI library loads and exports what include/fovraster.h declares, the
ctypes structs match the C layout, and the host wrappers keep the reference's error behaviour."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from tests.helpers import ROOT, syn
from fov3dgs_amd import _native


def _declared_functions():
    txt = open(os.path.join(ROOT, "include", "fovraster.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fr_[a-z_0-9A-Z]+)\s*\(", txt)) - {"fr_resize_fn"})


def test_library_exports_every_declared_symbol():
    lib = _native.load()
    names = _declared_functions()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_native.EXPORTS)
    assert lib.fr_abi_version() == _native.ABI_VERSION


def test_ctypes_structs_match_c_layout(tmp_path):
    src = tmp_path / "layout.c"
    fields_f = [f[0] for f in _native.ForwardArgs._fields_]
    fields_b = [f[0] for f in _native.BackwardArgs._fields_]
    body = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{ROOT}/include/fovraster.h"', 'int main(){',
            'printf("%zu\\n", sizeof(fr_forward_args));']
    body += [f'printf("%zu\\n", offsetof(fr_forward_args, {f}));' for f in fields_f]
    body += ['printf("%zu\\n", sizeof(fr_backward_args));']
    body += [f'printf("%zu\\n", offsetof(fr_backward_args, {f}));' for f in fields_b]
    body += ['return 0;}']
    src.write_text("\n".join(body))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-o", str(exe), str(src)])
    nums = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert nums[0] == C.sizeof(_native.ForwardArgs)
    for f, off in zip(fields_f, nums[1:1 + len(fields_f)]):
        assert getattr(_native.ForwardArgs, f).offset == off, f
    k = 1 + len(fields_f)
    assert nums[k] == C.sizeof(_native.BackwardArgs)
    for f, off in zip(fields_b, nums[k + 1:]):
        assert getattr(_native.BackwardArgs, f).offset == off, f


def test_workspace_sizes_are_host_computable():
    lib = _native.load()
    g0 = lib.fr_geometry_bytes(0, 1000)
    g3 = lib.fr_geometry_bytes(3, 1000)
    g2 = lib.fr_geometry_bytes(2, 1000)
    assert g3 > g2 >= 1000 * (48 + 24) and g0 > g2  # RF keeps per-level colours, the training variants their backward rows
    assert lib.fr_image_bytes(0, 1920, 1080) >= 1920 * 1080 * 8
    assert lib.fr_binning_bytes(0, 10) >= 120 and lib.fr_binning_bytes(0, 0) >= 0


def test_invalid_arguments_are_reported_not_executed():
    lib = _native.load()
    a = _native.ForwardArgs()
    a.variant = 8
    assert lib.fr_forward(C.byref(a)) == -1 and b"variant" in lib.fr_last_error()
    a.variant, a.P, a.W, a.H = 0, 5, 0, 16
    assert lib.fr_forward(C.byref(a)) == -1
    b = _native.BackwardArgs()
    b.variant = 3
    assert lib.fr_backward(C.byref(b)) == -1 and b"backward exists only" in lib.fr_last_error()


def test_rasterizer_argument_validation_and_no_cpu_fallback():
    from fov3dgs_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from fov3dgs_amd.diff_gaussian_rasterization_fov_pcheck_obb import GaussianRasterizer as FovRasterizer
    cloud, cam = syn.scene_1k(P=10), syn.camera_1k(32, 32)
    rs = GaussianRasterizationSettings(32, 32, 0.5, 0.5, torch.zeros(3), 1.0, cam.world_view_transform,
                                       cam.full_proj_transform, 3, cam.camera_center, False, False)
    r = GaussianRasterizer(rs)
    m2 = torch.zeros(10, 3)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(cloud.get_xyz, m2, cloud.get_opacity, scales=cloud.get_scaling, rotations=cloud.get_rotation)
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair"):
        r(cloud.get_xyz, m2, cloud.get_opacity, shs=cloud.get_features)
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair"):
        FovRasterizer(rs)(cloud.get_xyz, m2, cloud.get_opacity, shs_rest=cloud.get_rest_features)
    # CPU tensors: loud failure, never a silent CPU path
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        r(cloud.get_xyz, m2, cloud.get_opacity, shs=cloud.get_features, scales=cloud.get_scaling,
          rotations=cloud.get_rotation)


def test_cuda_type_dispatch():
    from fov3dgs_amd.gaussian_wrapper import get_gs_rasterizer
    from fov3dgs_amd import diff_gaussian_rasterization as r0, diff_gaussian_rasterization_pcheck_obb_sum as rs
    assert isinstance(get_gs_rasterizer("original", None), r0.GaussianRasterizer)
    assert isinstance(get_gs_rasterizer("pcheck_obb_sum", None), rs.GaussianRasterizer)
    with pytest.raises(ValueError, match="Invalid cuda type"):
        get_gs_rasterizer("nope", None)


def test_synthetic_scenes_are_deterministic():
    a, b = syn.scene_1k(), syn.scene_1k()
    assert torch.equal(a.get_xyz, b.get_xyz) and a.get_features.shape == (1000, 16, 3)
    big = syn.scene_bicycle_scale(P=20000, seed=1)
    assert big.get_xyz.shape == (20000, 3) and big.get_rotation.norm(dim=1).allclose(torch.ones(20000))
    hl, dcs, op = syn.foveation_layers(big)
    frac = [(hl == i).float().mean().item() for i in range(4)]
    np.testing.assert_allclose(frac, syn.LEVEL_FRACTIONS, atol=0.02)
    assert dcs.shape == (20000, 4, 3) and op.shape == (20000, 4) and (op >= 0).all() and (op <= 1).all()
    cam = syn.camera_ring(3)
    assert cam.image_width == 1920 and cam.image_height == 1080


def test_gradient_check_is_relative_row_by_row():
    """tests/checks.py: the gradient tolerance is relative per row (no floor tied to the tensor's largest entry): a 1 % error
    in the degree-3 SH coefficients' gradients (slots 9..15) of rows a million times smaller than the largest one fails."""
    from tests.checks import check_grad, grad_stats
    rng = np.random.default_rng(0)
    P = 20000
    # rows spanning eight orders of magnitude, as dL_dsh of a real frame does
    want = rng.normal(size=(P, 16, 3)) * (10.0 ** rng.uniform(-6, 2, size=(P, 1, 1)))
    want = want.astype(np.float32)
    noise = (want.astype(np.float64) * (1 + 3e-7 * rng.normal(size=want.shape))).astype(np.float32)  # fp32 rounding noise
    check_grad(noise, want, "rounding noise passes")
    spoiled = want.copy()
    spoiled[:, 9:, :] *= 1.01
    st = grad_stats(spoiled, want)
    assert st["frac_bad"] > 0.99
    with pytest.raises(AssertionError):
        check_grad(spoiled, want, "1 % in coefficients 9..15")
    # ... and only on the small rows (what round 2's absolute floor, 1e-5 x the tensor's maximum, let through)
    small = np.abs(want).reshape(P, -1).max(axis=1) < 1e-3 * np.abs(want).max()
    spoiled = want.copy()
    spoiled[small, 9:, :] *= 1.01
    with pytest.raises(AssertionError):
        check_grad(spoiled, want, "1 % in the small rows only")
    # a few flipped rows are within the outlier budget, but are recorded
    few = want.copy()
    few[:5] *= 1.5
    assert 0 < grad_stats(few, want)["frac_bad"] <= 1e-3


def test_cuda_type_table_matches_the_reference_strings():
    from fov3dgs_amd.gaussian_wrapper import rasterizer_class
    for name in ("original", "pcheck_obb", "pcheck_obb_max", "pcheck_obb_sum", "pcheck_obb_loss_weighted_max_count"):
        assert rasterizer_class(name).__name__ == "GaussianRasterizer"


def test_fast_path_recognises_the_reference_getters_only():
    """gaussian_renderer._getter_fingerprint_ok: the raw-parameter fast path of render() is taken only for classes whose
    getters are the reference's one-liners (scene/gaussian_model.py:200-240) as resolved through the MRO; a subclass that overrides one
    getter, adds a factor or a clamp, or a wrapper with other code keeps its getters."""
    import torch
    from fov3dgs_amd import gaussian_renderer as gr
    from fov3dgs_amd import synthetic as syn

    class Clamped(syn.ReferenceShapedModel):
        @property
        def get_scaling(self):
            return self.scaling_activation(self._scaling).clamp(max=1.0)

    class Doubled(syn.ReferenceShapedModel):
        @property
        def get_opacity(self):
            return 0.5 * self.opacity_activation(self._opacity)

    class OtherConcat(syn.ReferenceShapedModel):
        @property
        def get_features(self):
            return torch.cat((self._features_dc, self._features_rest), dim=2)

    class Plain(syn.ReferenceShapedModel):
        """A subclass that overrides nothing render() bypasses."""
        def extra(self):
            return 1

    class NotAProperty(syn.ReferenceShapedModel):
        def get_rotation(self):
            return self.rotation_activation(self._rotation)
    assert gr._getter_fingerprint_ok(syn.ReferenceShapedModel) and gr._getter_fingerprint_ok(Plain)
    for cls in (Clamped, Doubled, OtherConcat, NotAProperty, syn.MaskedOpacityModel, syn.ReferenceGetterModel, syn.GaussianCloud):
        assert not gr._getter_fingerprint_ok(cls), cls.__name__
    # CPU tensors never take the fast path (and nothing is rendered on the CPU)
    assert gr._reference_model_fields(syn.ReferenceShapedModel(syn.scene_1k(P=10))) is None
