"""`cuda_type` string -> rasterizer module of that variant (the boundary the reference exposes as
fov3dgs/gaussian_wrapper.py:11-23 `get_gs_rasterizer`; same strings, same ValueError for an unknown one)."""
import importlib

_MODULES = {
    "original": "diff_gaussian_rasterization",
    "pcheck_obb": "diff_gaussian_rasterization_pcheck_obb",
    "pcheck_obb_max": "diff_gaussian_rasterization_pcheck_obb_max",
    "pcheck_obb_sum": "diff_gaussian_rasterization_pcheck_obb_sum",
    "pcheck_obb_loss_weighted_max_count": "diff_gaussian_rasterization_pcheck_obb_loss_weighted_max_count",
}


def rasterizer_class(cuda_type):
    """The GaussianRasterizer class behind a `cuda_type` string."""
    name = _MODULES.get(cuda_type)
    if name is None:
        raise ValueError("Invalid cuda type: {}".format(cuda_type))
    return importlib.import_module("." + name, __package__).GaussianRasterizer


def get_gs_rasterizer(cuda_type, raster_settings):
    return rasterizer_class(cuda_type)(raster_settings=raster_settings)
