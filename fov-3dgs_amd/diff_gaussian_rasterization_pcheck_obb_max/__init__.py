"""Drop-in for the reference package fov3dgs/submodules/diff-gaussian-rasterization_pcheck_obb_max/…/__init__.py (pruning metric "max_contrib": gaussians_count per in-support pixel, contributions = max alpha*T).

Same public names: GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians.
"""
from .. import _native
from ..rasterizer import GaussianRasterizationSettings, _make_plain  # noqa: F401

_RasterizeGaussians, rasterize_gaussians, GaussianRasterizer = _make_plain(
    _native.VARIANT_PCHECK_OBB_MAX, with_counts=True, has_backward=True)
