"""Drop-in for the reference package fov3dgs/submodules/diff-gaussian-rasterization_pcheck_obb/…/__init__.py (inference rasterizer used by render.py).

Same public names: GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians.
"""
from .. import _native
from ..rasterizer import GaussianRasterizationSettings, _make_plain  # noqa: F401

_RasterizeGaussians, rasterize_gaussians, GaussianRasterizer = _make_plain(
    _native.VARIANT_PCHECK_OBB, with_counts=False, has_backward=False)
