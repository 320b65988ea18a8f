"""Run the HIP library on oracle-style input dicts and pull its internal state back as numpy."""
import numpy as np
import torch

from fov3dgs_amd import _native
from fov3dgs_amd.rasterizer import GaussianRasterizationSettings, _backward_native, _forward_native

VARIANT_IDS = _native.VARIANT_IDS


def _t(x, dev):
    return None if x is None else torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32).to(dev)


def settings_from(cam, dev, debug=True):
    return GaussianRasterizationSettings(
        image_height=int(cam["image_height"]), image_width=int(cam["image_width"]), tanfovx=float(cam["tanfovx"]),
        tanfovy=float(cam["tanfovy"]), bg=_t(cam["bg"], dev), scale_modifier=float(cam.get("scale_modifier", 1.0)),
        viewmatrix=_t(cam["viewmatrix"], dev), projmatrix=_t(cam["projmatrix"], dev), sh_degree=int(cam["sh_degree"]),
        campos=_t(cam["campos"], dev), prefiltered=False, debug=debug)


def _view(buf, ptr, count, dtype):
    off = ptr - buf.data_ptr()
    nbytes = count * torch.empty(0, dtype=dtype).element_size()
    assert 0 <= off and off + nbytes <= buf.numel()
    return buf[off:off + nbytes].view(dtype)


def vis_list_of(lib, vid, P, geom):
    """The library's list of cull survivors (Gaussian indices, increasing) as an int64 device tensor."""
    n = int(_view(geom, lib.fr_geometry_vis_count(vid, P, geom.data_ptr()), 1, torch.int32).item())
    return _view(geom, lib.fr_geometry_vis_list(vid, P, geom.data_ptr()), n, torch.int32).long()


def hip_forward(variant, scene, cam, dev="cuda:0", debug=True, packed=False):
    """-> dict shaped like oracle.forward()'s (the subset the HIP library keeps).
    packed: also hand the model over in the packed static-model layout (needs scales + rotations + 16 SH coefficients)."""
    lib = _native.load()
    vid = VARIANT_IDS[variant]
    rs = settings_from(cam, dev, debug)
    tens = {k: _t(scene.get(k), dev) for k in ("means3D", "shs", "colors_precomp", "opacities", "scales", "rotations",
                                                "cov3D_precomp", "shs_dcs", "highest_levels")}
    pk = None
    if packed:
        from fov3dgs_amd.rasterizer import pack_model
        pk = pack_model(tens["means3D"], tens["scales"], tens["rotations"], tens["opacities"], shs=tens["shs"],
                        shs_dcs=tens["shs_dcs"], highest_levels=tens["highest_levels"])
    res = _forward_native(vid, rs, tens["means3D"], tens["shs"], tens["colors_precomp"], tens["opacities"],
                          tens["scales"], tens["rotations"], tens["cov3D_precomp"], tens["shs_dcs"],
                          tens["highest_levels"], cam.get("gaze", (0.5, 0.5)), cam.get("alpha", 0.05),
                          loss_map=_t(scene.get("loss_map"), dev), packed=pk, cur_level=cam.get("cur_level", 0.0))
    torch.cuda.synchronize()
    num_rendered, color, radii, geom, binb, img = res[:6]
    W, H = rs.image_width, rs.image_height
    T = ((W + 15) // 16) * ((H + 15) // 16)
    out = {"num_rendered": num_rendered, "color": color.cpu().numpy(), "radii": radii.cpu().numpy(),
           "_tensors": tens, "_rs": rs, "_buffers": (geom, binb, img), "_radii_t": radii,
           "_lease": res[-1]}  # the workspace set stays reserved while this dict lives
    if img.numel() == 0 or tens["means3D"].shape[0] == 0:  # P == 0: the library returns before touching any workspace
        out["ranges"], out["point_list"] = np.zeros((T, 2), np.uint32), np.zeros(0, np.uint32)
        return out
    rptr = lib.fr_image_ranges(vid, W, H, img.data_ptr())
    out["ranges"] = _view(img, rptr, 2 * T, torch.int32).cpu().numpy().astype(np.uint32).reshape(T, 2)
    P = tens["means3D"].shape[0]
    vis = vis_list_of(lib, vid, P, geom)
    out["vis_list"] = vis.cpu().numpy().astype(np.uint32)
    if num_rendered > 0:
        # the library's per-tile lists hold ITEMS (positions in its index-ordered list of cull survivors, vis_list): the
        # reference's point_list is their Gaussian indices
        pptr = lib.fr_binning_point_list(vid, num_rendered, binb.data_ptr())
        items = _view(binb, pptr, num_rendered, torch.int32).long()
        out["point_list"] = vis[items].cpu().numpy().astype(np.uint32)
    else:
        out["point_list"] = np.zeros(0, np.uint32)
    if variant in ("original", "pcheck_obb_sum", "pcheck_obb_max", "pcheck_obb_loss_weighted_max_count"):
        out["final_T"] = _view(img, lib.fr_image_final_T(vid, W, H, img.data_ptr()), W * H, torch.float32).cpu().numpy().reshape(H, W)
        out["n_contrib"] = _view(img, lib.fr_image_n_contrib(vid, W, H, img.data_ptr()), W * H, torch.int32).cpu().numpy().astype(np.uint32).reshape(H, W)
    if variant in ("pcheck_obb_sum", "pcheck_obb_max", "pcheck_obb_loss_weighted_max_count"):
        out["gaussians_count"], out["contributions"] = res[6].cpu().numpy(), res[7].cpu().numpy()
    if variant in ("fov_pcheck_obb", "naive_pcheck_obb"):
        lv = _view(img, lib.fr_image_tile_levels(W, H, img.data_ptr()), 5 * T, torch.float32).cpu().numpy().reshape(5, T)
        out["tile_levels"], out["tile_min"], out["tile_gx"], out["tile_gy"] = lv[0], lv[1], lv[2], lv[3]
        out["tile_blend"] = (lv[4] != 0).astype(np.uint8)
    return out


def hip_backward(variant, fwd, dL_dpix, dev="cuda:0"):
    vid = VARIANT_IDS[variant]
    t, rs = fwd["_tensors"], fwd["_rs"]
    geom, binb, img = fwd["_buffers"]
    empty = torch.Tensor([])
    g = _backward_native(vid, rs, t["means3D"], fwd["_radii_t"], t["colors_precomp"] if t["colors_precomp"] is not None else empty,
                         t["opacities"], t["scales"] if t["scales"] is not None else empty,
                         t["rotations"] if t["rotations"] is not None else empty,
                         t["cov3D_precomp"] if t["cov3D_precomp"] is not None else empty, _t(dL_dpix, dev),
                         t["shs"] if t["shs"] is not None else empty, geom, fwd["num_rendered"], binb, img, want_cov3D_grad=True, want_color_grad=True)
    torch.cuda.synchronize()
    names = ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot")
    return {n: v.cpu().numpy() for n, v in zip(names, g)}
