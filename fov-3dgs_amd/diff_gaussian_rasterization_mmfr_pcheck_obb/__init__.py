"""Drop-in for the reference package
fov3dgs/submodules/diff-gaussian-rasterization_mmfr_pcheck_obb/diff_gaussian_rasterization_mmfr_pcheck_obb/__init__.py
(one level of the multi-model foveated baseline "MMFR" behind gaussian_renderer_fov_mmfr.render(); fps/MMFR-Q in the
paper's table).
"""
from ..rasterizer import GaussianRasterizationSettings, _make_mmfr  # noqa: F401

_RasterizeGaussians, rasterize_gaussians, GaussianRasterizer = _make_mmfr()
