"""Drop-in for the reference package fov3dgs/submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py (cuda_type "original").

Same public names: GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians.
"""
from .. import _native
from ..rasterizer import GaussianRasterizationSettings, _make_plain  # noqa: F401

_RasterizeGaussians, rasterize_gaussians, GaussianRasterizer = _make_plain(
    _native.VARIANT_ORIGINAL, with_counts=False, has_backward=True)
