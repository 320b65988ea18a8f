"""cuda_type -> rasterizer variant (reference: fov3dgs/gaussian_wrapper.py:11-23)."""
from .diff_gaussian_rasterization import GaussianRasterizer as GaussianRasterizer_original
from .diff_gaussian_rasterization_pcheck_obb import GaussianRasterizer as GaussianRasterizer_pcheck_obb
from .diff_gaussian_rasterization_pcheck_obb_sum import GaussianRasterizer as GaussianRasterizer_pcheck_obb_sum


def get_gs_rasterizer(cuda_type, raster_settings):
    if cuda_type == "original":
        return GaussianRasterizer_original(raster_settings=raster_settings)
    elif cuda_type == "pcheck_obb":
        return GaussianRasterizer_pcheck_obb(raster_settings=raster_settings)
    elif cuda_type == "pcheck_obb_sum":
        return GaussianRasterizer_pcheck_obb_sum(raster_settings=raster_settings)
    elif cuda_type in ("pcheck_obb_max", "pcheck_obb_loss_weighted_max_count"):
        # pruning-metric variants of the reference: not built yet
        raise NotImplementedError("cuda type {} is not implemented in the HIP library yet".format(cuda_type))
    else:
        raise ValueError("Invalid cuda type: {}".format(cuda_type))
