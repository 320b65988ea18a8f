"""Drop-in for the reference package
fov3dgs/submodules/diff-gaussian-rasterization_fov_pcheck_obb/diff_gaussian_rasterization_fov_pcheck_obb/__init__.py
(the foveated, inference-only rasterizer behind gaussian_renderer_fov.render()).
"""
from ..rasterizer import GaussianRasterizationSettings, _make_fov  # noqa: F401

_RasterizeGaussians, rasterize_gaussians, GaussianRasterizer = _make_fov()
