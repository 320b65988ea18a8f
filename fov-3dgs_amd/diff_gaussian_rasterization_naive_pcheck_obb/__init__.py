"""Drop-in for the reference package
fov3dgs/submodules/diff-gaussian-rasterization_naive_pcheck_obb/diff_gaussian_rasterization_naive_pcheck_obb/__init__.py
(the shared-model foveated baseline "SMFR" behind gaussian_renderer_fov_naive.render(); fps/naiveFR in the paper's table).
"""
from ..rasterizer import GaussianRasterizationSettings, _make_naive_fov  # noqa: F401

_RasterizeGaussians, rasterize_gaussians, GaussianRasterizer = _make_naive_fov()
