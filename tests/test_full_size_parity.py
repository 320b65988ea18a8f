"""GPU parity at BASELINE's full size: the S-6M cloud (6 000 000 Gaussians) at 1920x1080 against the CPU oracle.

The oracle preprocesses, culls and walks ALL 6 M Gaussians (so `radii`, which the OBB / foveal cull resets, is
compared bit for bit over the whole cloud -- this is what proves that the cull pass's conservative frame test and the
clipped walks never drop a Gaussian the reference keeps), and bins / sorts / blends a window of tiles
(`orc_in.win`), inside which instance lists are compared bit for bit and pixels / gradients within the tolerances of
tests/checks.py. Backward: dL_dpix is zero outside the window, so that every per-Gaussian gradient sum only has
terms from window pixels and the oracle's sums over the window are the whole answer.

BASELINE configs covered: 2 (non-foveated forward, pcheck_obb), 3 (4-layer foveated, centred + moving gaze, packed
and ordinary model layout), 4 (training step: pcheck_obb_sum forward statistics + backward gradients).
Reference: RF rasterizer_impl.cu:264-383 (filter), RF forward.cu:262-609 (blend), RS forward.cu:298-430, R0 backward.cu.
"""
import math
import os

import numpy as np
import pytest
import torch

from tests.checks import check_grad, check_image
from tests.helpers import cam_dict, scene_dict, syn
from tests import parity_report
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

W, H = 1920, 1080
GX, GY = 120, 68
T = GX * GY


class S6M:
    """The bench scene, built once per session: CPU cloud (oracle inputs) + device tensors (rasterizer inputs)."""

    def __init__(self):
        assert torch.cuda.is_available(), "no GPU visible: -m gpu tests must run on the MI355X box"
        from fov3dgs_amd import _native, rasterizer as rz
        self.rz, self.native, self.lib = rz, _native, _native.load()
        self.dev = torch.device("cuda", 0)
        orc.set_threads(os.cpu_count() or 1)
        self.cloud = syn.scene_bicycle_scale(P=6_000_000, seed=1)
        self.fov = syn.foveation_layers(self.cloud, seed=2)
        self.cam = syn.camera_ring(0, 8, W, H)  # the bench's camera at N = 1
        self.scene_plain = scene_dict(self.cloud, "pcheck_obb")
        self.scene_fov = scene_dict(self.cloud, "fov_pcheck_obb", self.fov)
        # the rasterizer gets the very arrays the oracle gets (activations evaluated once, on the CPU: torch's GPU exp /
        # sigmoid / normalize differ from the CPU's in the last bit for a fifth of the elements, and the covariance's
        # eigenvalues amplify that to hundreds of ulps)
        up = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(self.dev)
        sp, sf = self.scene_plain, self.scene_fov
        self.xyz, self.sc, self.rot = up(sp["means3D"]), up(sp["scales"]), up(sp["rotations"])
        self.opac, self.sh, self.rest = up(sp["opacities"]), up(sp["shs"]), up(sf["shs"])
        self.highest, self.shs_dcs, self.opac4 = up(sf["highest_levels"]), up(sf["shs_dcs"]), up(sf["opacities"])
        self.cam_dev = syn.camera_ring(0, 8, W, H).to(self.dev)
        self.bg = (0.05, 0.1, 0.15)
        c = self.cam_dev
        self.rs = rz.GaussianRasterizationSettings(H, W, math.tan(c.FoVx * 0.5), math.tan(c.FoVy * 0.5),
                                                   torch.tensor(self.bg, device=self.dev), 1.0, c.world_view_transform,
                                                   c.full_proj_transform, 3, c.camera_center, False, False)
        self.packed_fov = self.packed_plain = None

    def cam_dict(self, gaze=(0.5, 0.5), window=None):
        cd = cam_dict(self.cam, bg=self.bg, gaze=gaze, alpha=0.05)
        cd["tile_window"] = window
        cd["capacity_hint"] = 12_000_000
        return cd

    def hip(self, variant, gaze=(0.5, 0.5), packed=False):
        """-> dict of numpy / tensors from one native forward call of `variant` on the resident scene."""
        rz, vid, E = self.rz, self.native.VARIANT_IDS[variant], torch.Tensor([])
        pk = None
        with torch.no_grad():
            if variant == "fov_pcheck_obb":
                if packed:
                    if self.packed_fov is None:
                        self.packed_fov = rz.pack_model(self.xyz, self.sc, self.rot, self.opac4, shs=self.rest, shs_dcs=self.shs_dcs,
                                                        highest_levels=self.highest)
                    pk = self.packed_fov
                res = rz._forward_native(vid, self.rs, self.xyz, self.rest, E, self.opac4, self.sc, self.rot, E, self.shs_dcs,
                                         self.highest, gaze, 0.05, packed=pk)
            else:
                if packed:
                    if self.packed_plain is None:
                        self.packed_plain = rz.pack_model(self.xyz, self.sc, self.rot, self.opac, shs=self.sh)
                    pk = self.packed_plain
                res = rz._forward_native(vid, self.rs, self.xyz, self.sh, E, self.opac, self.sc, self.rot, E, packed=pk)
            torch.cuda.synchronize()
        D, color, radii, geom, binb, img = res[:6]

        def view(buf, ptr, count, dtype):
            off = ptr - buf.data_ptr()
            return buf[off:off + 4 * count].view(dtype)
        out = dict(num_rendered=D, color=color, radii=radii, buffers=(geom, binb, img), lease=res[-1], res=res)
        out["ranges"] = view(img, self.lib.fr_image_ranges(vid, W, H, img.data_ptr()), 2 * T, torch.int32).view(T, 2).long()
        out["point_list"] = view(binb, self.lib.fr_binning_point_list(vid, D, binb.data_ptr()), D, torch.int32)
        if variant == "pcheck_obb_sum":
            out["final_T"] = view(img, self.lib.fr_image_final_T(vid, W, H, img.data_ptr()), W * H, torch.float32).view(H, W)
            out["n_contrib"] = view(img, self.lib.fr_image_n_contrib(vid, W, H, img.data_ptr()), W * H, torch.int32).view(H, W)
            out["gaussians_count"], out["contributions"] = res[6], res[7]
        return out


@pytest.fixture(scope="module")
def s6m():
    return S6M()


def window_tiles(win):
    x0, y0, x1, y1 = win
    ty, tx = np.meshgrid(np.arange(y0, y1), np.arange(x0, x1), indexing="ij")
    return (ty * GX + tx).reshape(-1)


def compare_lists(got, want, win, tag):
    """Instance lists of every tile of the window, bit for bit (order included)."""
    tiles = window_tiles(win)
    g_rng = got["ranges"].cpu().numpy()
    w_rng = want["ranges"].astype(np.int64)
    g_len = g_rng[tiles, 1] - g_rng[tiles, 0]
    w_len = w_rng[tiles, 1] - w_rng[tiles, 0]
    np.testing.assert_array_equal(g_len, w_len, err_msg=tag + ": per-tile instance counts")
    assert int(w_len.sum()) == want["num_rendered"] and int(w_len.sum()) > 100_000, tag
    # gather both sides in (window tile, position) order
    rep = np.repeat(np.arange(len(tiles)), w_len)
    pos = np.arange(int(w_len.sum())) - np.repeat(np.cumsum(w_len) - w_len, w_len)
    gi = torch.as_tensor(g_rng[tiles, 0][rep] + pos, device=got["point_list"].device)
    g_ids = got["point_list"][gi].cpu().numpy().astype(np.uint32)
    w_ids = want["point_list"][w_rng[tiles, 0][rep] + pos]
    np.testing.assert_array_equal(g_ids, w_ids, err_msg=tag + ": sorted instance lists")
    return int(w_len.sum()), int(w_len.max())


def crop(img, win):
    x0, y0, x1, y1 = win
    return img[..., y0 * 16:min(y1 * 16, H), x0 * 16:min(x1 * 16, W)]


CENTRE_WIN = (44, 18, 76, 50)  # 32 x 32 tiles around the image centre


def gaze_window(gaze, rows=9):
    """A full-width strip of `rows` tile rows through the gaze point: it crosses all four eccentricity levels and
    ~200 two-level tiles (a square window around the gaze only holds levels 0 and 1)."""
    cy = int(gaze[1] * GY)
    y0 = min(max(cy - rows // 2, 0), GY - rows)
    return (0, y0, GX, y0 + rows)


def test_plain_forward_full_size(s6m):
    """Config 2: pcheck_obb over the whole S-6M cloud -- radii of all 6 M Gaussians, lists + pixels of the window."""
    want = orc.forward("pcheck_obb", s6m.scene_plain, s6m.cam_dict(window=CENTRE_WIN))
    for packed in (False, True):
        got = s6m.hip("pcheck_obb", packed=packed)
        tag = f"pcheck_obb S-6M packed={packed}"
        np.testing.assert_array_equal(got["radii"].cpu().numpy(), want["radii"], err_msg=tag + ": radii over all Gaussians")
        n, longest = compare_lists(got, want, CENTRE_WIN, tag)
        check_image(crop(got["color"], CENTRE_WIN).cpu().numpy(), crop(want["color"], CENTRE_WIN), name=tag)
        parity_report.record("lists", tag, gaussians=int(s6m.xyz.shape[0]), visible=int((want["radii"] > 0).sum()),
                             window_instances=n, longest_window_list=longest, frame_instances=int(got["num_rendered"]))
    s6m.plain_radii = want["radii"]


@pytest.mark.parametrize("gaze_id", ("centre", "lissajous10", "lissajous47"))
def test_foveated_forward_full_size(s6m, gaze_id):
    """Config 3: fov_pcheck_obb, centred gaze and two gazes of the bench's Lissajous path; ordinary and packed model."""
    gaze = (0.5, 0.5) if gaze_id == "centre" else syn.lissajous_gaze(int(gaze_id[9:]), 90)
    win = gaze_window(gaze)
    want = orc.forward("fov_pcheck_obb", s6m.scene_fov, s6m.cam_dict(gaze=gaze, window=win))
    tiles = window_tiles(win)
    assert want["tile_blend"][tiles].sum() > 100 and len(np.unique(want["tile_min"][tiles].astype(int))) == 4, \
        "the window should cross all four levels and hold two-level tiles"
    for packed in (False, True):
        got = s6m.hip("fov_pcheck_obb", gaze=gaze, packed=packed)
        tag = f"fov_pcheck_obb S-6M gaze={gaze_id} packed={packed}"
        np.testing.assert_array_equal(got["radii"].cpu().numpy(), want["radii"], err_msg=tag + ": radii over all Gaussians")
        n, longest = compare_lists(got, want, win, tag)
        check_image(crop(got["color"], win).cpu().numpy(), crop(want["color"], win), name=tag)
        parity_report.record("lists", tag, visible=int((want["radii"] > 0).sum()), window_instances=n, longest_window_list=longest,
                             frame_instances=int(got["num_rendered"]), two_level_tiles_in_window=int(want["tile_blend"][tiles].sum()))


def test_shared_model_baseline_full_size(s6m):
    """SURVEY 8f rank 4 at full size: the SMFR baseline on the S-6M cloud (plain model + the foveated model's highest levels)."""
    gaze = syn.lissajous_gaze(10, 90)
    win = gaze_window(gaze)
    scene = dict(s6m.scene_plain, highest_levels=s6m.scene_fov["highest_levels"])
    want = orc.forward("naive_pcheck_obb", scene, s6m.cam_dict(gaze=gaze, window=win))
    rz, E = s6m.rz, torch.Tensor([])
    with torch.no_grad():
        res = rz._forward_native(s6m.native.VARIANT_IDS["naive_pcheck_obb"], s6m.rs, s6m.xyz, s6m.sh, E, s6m.opac, s6m.sc, s6m.rot, E, None,
                                 s6m.highest, gaze, 0.05)
        torch.cuda.synchronize()
    vid = s6m.native.VARIANT_IDS["naive_pcheck_obb"]
    D, color, radii, geom, binb, img = res[:6]
    view = lambda buf, ptr, count, dtype: buf[ptr - buf.data_ptr():ptr - buf.data_ptr() + 4 * count].view(dtype)
    got = dict(num_rendered=D, ranges=view(img, s6m.lib.fr_image_ranges(vid, W, H, img.data_ptr()), 2 * T, torch.int32).view(T, 2).long(),
               point_list=view(binb, s6m.lib.fr_binning_point_list(vid, D, binb.data_ptr()), D, torch.int32))
    tag = "naive_pcheck_obb (SMFR) S-6M"
    np.testing.assert_array_equal(radii.cpu().numpy(), want["radii"], err_msg=tag + ": radii over all Gaussians")
    compare_lists(got, want, win, tag)
    check_image(crop(color, win).cpu().numpy(), crop(want["color"], win), name=tag)


BWD_WIN = (30, 20, 90, 48)  # 60 x 28 tiles


def test_training_step_full_size(s6m):
    """Config 4: pcheck_obb_sum forward statistics and the backward pass on the S-6M cloud."""
    from fov3dgs_amd.rasterizer import _backward_native
    win = BWD_WIN
    want = orc.forward("pcheck_obb_sum", s6m.scene_plain, s6m.cam_dict(window=win))
    got = s6m.hip("pcheck_obb_sum")
    tag = "pcheck_obb_sum S-6M"
    np.testing.assert_array_equal(got["radii"].cpu().numpy(), want["radii"], err_msg=tag + ": radii")
    compare_lists(got, want, win, tag)
    check_image(crop(got["color"], win).cpu().numpy(), crop(want["color"], win), name=tag)
    g_nc, w_nc = crop(got["n_contrib"], win).cpu().numpy().astype(np.uint32), crop(want["n_contrib"], win)
    same = g_nc == w_nc
    parity_report.record("count", tag + " n_contrib", frac_differ=float(np.mean(~same)))
    assert np.mean(~same) <= 1e-3
    np.testing.assert_allclose(crop(got["final_T"], win).cpu().numpy()[same], crop(want["final_T"], win)[same], rtol=1e-4, atol=1e-7)
    # backward: random dL_dpix inside the window, zero outside
    x0, y0, x1, y1 = win
    dpix = np.zeros((3, H, W), np.float32)
    dpix[:, y0 * 16:y1 * 16, x0 * 16:x1 * 16] = np.random.default_rng(3).normal(size=(3, (y1 - y0) * 16, (x1 - x0) * 16))
    wg = orc.backward("pcheck_obb_sum", s6m.scene_plain, s6m.cam_dict(window=win), want, dpix)
    geom, binb, img = got["buffers"]
    E = torch.Tensor([])
    res = _backward_native(s6m.native.VARIANT_IDS["pcheck_obb_sum"], s6m.rs, s6m.xyz, got["radii"], E, s6m.opac, s6m.sc, s6m.rot, E,
                           torch.as_tensor(dpix, device=s6m.dev), s6m.sh, geom, got["num_rendered"], binb, img, want_cov3D_grad=True, want_color_grad=True)
    torch.cuda.synchronize()
    names = ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot")
    touched = None
    for k, v in zip(names, res):
        g = v.cpu().numpy().reshape(wg[k].shape)
        # rows the window reaches: compare those (the rest must be exactly zero on both sides)
        rows = np.abs(wg[k]).reshape(len(g), -1).max(axis=1) > 0
        grows = np.abs(g).reshape(len(g), -1).max(axis=1) > 0
        if k == "dL_dopacity":
            touched = rows
        assert not (grows & ~rows).any() or np.abs(g[grows & ~rows]).max() < 1e-6, k + ": gradient on a Gaussian the window cannot reach"
        check_grad(g[rows], wg[k][rows], f"{tag} {k}")
    assert touched.sum() > 10_000
    parity_report.record("count", tag + " Gaussians with a gradient from the window", n=int(touched.sum()))


def test_raw_parameters_full_size():
    """fr_forward_args.raw_activations on the S-6M cloud at 1080p: the training variant fed with the model's raw parameters
    gives the image, radii and statistics of the activate-then-render path bit for bit, and the same gradients (w.r.t. the
    raw parameters) up to the order of the float atomics."""
    from fov3dgs_amd.activations import activate
    from fov3dgs_amd.diff_gaussian_rasterization_pcheck_obb_sum import GaussianRasterizationSettings, GaussianRasterizer
    dev = "cuda:0"
    cam = syn.camera_ring(0, 8).to(dev)
    W, H = cam.image_width, cam.image_height
    rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=dev), 1.0,
                                       cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
    w = torch.rand(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(3)) - 0.5
    res = []
    for raw in (False, True):
        cloud = syn.scene_bicycle_scale(P=6_000_000, seed=1).to(dev).requires_grad_(True)
        s, q, o = (cloud._scaling, cloud._rotation, cloud._opacity) if raw else activate(cloud._scaling, cloud._rotation, cloud._opacity)
        out = GaussianRasterizer(rs)(means3D=cloud.get_xyz, means2D=torch.zeros_like(cloud.get_xyz, requires_grad=True), opacities=o,
                                     shs=cloud.get_features_split, scales=s, rotations=q, **({"raw_activations": True} if raw else {}))
        (out[0] * w).sum().backward()
        res.append(dict(img=out[0].detach(), radii=out[1], count=out[2], contrib=out[3].detach(),
                        grads={n: getattr(cloud, "_" + n).grad for n in ("xyz", "scaling", "rotation", "opacity", "features_dc", "features_rest")}))
        del cloud, out, s, q, o
    a, b = res
    assert torch.equal(a["img"], b["img"]) and torch.equal(a["radii"], b["radii"]) and torch.equal(a["count"], b["count"])
    assert int((a["radii"] > 0).sum()) > 1_000_000
    tol = float((a["contrib"] - b["contrib"]).abs().max()) / (float(a["contrib"].abs().max()) + 1e-12)
    assert tol < 1e-5, tol  # sums of float atomics
    for n in a["grads"]:
        check_grad(b["grads"][n].cpu().numpy(), a["grads"][n].cpu().numpy(), "S-6M raw parameters: " + n)
