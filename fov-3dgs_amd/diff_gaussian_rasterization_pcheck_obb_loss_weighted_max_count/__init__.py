"""Drop-in for the reference package fov3dgs/submodules/diff-gaussian-rasterization_pcheck_obb_loss_weighted_max_count/…/__init__.py (pruning metrics "max_comp_efficiency" / "surface": each pixel's loss_map value is credited to its max-contribution Gaussian; forward takes `loss_map`).

Same public names: GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians.
"""
from .. import _native
from ..rasterizer import GaussianRasterizationSettings, _make_plain  # noqa: F401

_RasterizeGaussians, rasterize_gaussians, GaussianRasterizer = _make_plain(
    _native.VARIANT_PCHECK_OBB_LWMC, with_counts=True, has_backward=True, takes_loss_map=True)
