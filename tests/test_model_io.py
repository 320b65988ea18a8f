"""3DGS PLY files and the composed per-level artefacts (fov-3dgs_amd/model_io.py).

The reference's own reader/writer (scene/gaussian_model.py:356-540, compose_models.py:41-80) needs `plyfile` and its
CUDA extensions to import, so these tests pin the FORMAT: a file laid out by hand exactly as save_ply_index writes it
(property order, channel-major SH, little-endian float32 + int32 index) must load to the expected tensors, and
compose_levels must reproduce the carry-forward / overwrite semantics on it.
"""
import os
import struct

import numpy as np
import pytest
import torch

from tests.helpers import syn
from fov3dgs_amd import model_io


def _hand_written_ply(path, P, with_index, rng):
    """The byte layout of GaussianModel.save_ply(_index), written without model_io."""
    names = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(45)] + \
            ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]
    vals = rng.normal(size=(P, len(names))).astype(np.float32)
    idx = rng.permutation(1000)[:P].astype(np.int32)
    with open(path, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\ncomment hand written\n")
        f.write(f"element vertex {P}\n".encode())
        for n in names:
            f.write(f"property float {n}\n".encode())
        if with_index:
            f.write(b"property int index\n")
        f.write(b"end_header\n")
        for r in range(P):
            f.write(struct.pack("<%df" % len(names), *vals[r]))
            if with_index:
                f.write(struct.pack("<i", int(idx[r])))
    return names, vals, idx


def test_load_ply_reads_the_reference_layout(tmp_path):
    rng = np.random.default_rng(5)
    path = str(tmp_path / "point_cloud.ply")
    names, vals, idx = _hand_written_ply(path, 37, True, rng)
    cloud, indexes = model_io.load_ply(path)
    col = {n: vals[:, i] for i, n in enumerate(names)}
    np.testing.assert_array_equal(cloud._xyz.numpy(), np.stack([col["x"], col["y"], col["z"]], 1))
    np.testing.assert_array_equal(cloud._opacity.numpy()[:, 0], col["opacity"])
    # f_dc_c -> features_dc[:, 0, c]; f_rest_{c*15 + k} -> features_rest[:, k, c]  (channel-major on disk)
    for c in range(3):
        np.testing.assert_array_equal(cloud._features_dc.numpy()[:, 0, c], col[f"f_dc_{c}"])
        for k in (0, 7, 14):
            np.testing.assert_array_equal(cloud._features_rest.numpy()[:, k, c], col[f"f_rest_{c * 15 + k}"])
    np.testing.assert_array_equal(cloud._scaling.numpy(), np.stack([col[f"scale_{i}"] for i in range(3)], 1))
    np.testing.assert_array_equal(cloud._rotation.numpy(), np.stack([col[f"rot_{i}"] for i in range(4)], 1))
    np.testing.assert_array_equal(indexes.numpy(), idx.astype(np.int64))
    assert cloud._features_rest.shape == (37, 15, 3) and cloud._features_dc.shape == (37, 1, 3)
    cloud2, none = model_io.load_ply(_write(tmp_path, False, rng))
    assert none is None and cloud2._xyz.shape == (11, 3)


def _write(tmp_path, with_index, rng):
    p = str(tmp_path / ("b_%d.ply" % with_index))
    _hand_written_ply(p, 11, with_index, rng)
    return p


def test_save_load_round_trip_and_header(tmp_path):
    cloud = syn.scene_1k(P=123, seed=3)
    idx = torch.arange(123, dtype=torch.int32).flip(0)
    path = str(tmp_path / "m" / "point_cloud.ply")
    model_io.save_ply(path, cloud, idx)
    raw = open(path, "rb").read()
    header = raw[:raw.index(b"end_header\n")].decode().split("\n")
    props = [l.split()[-1] for l in header if l.startswith("property")]
    assert props[:9] == ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
    assert props[9:54] == [f"f_rest_{i}" for i in range(45)] and props[54] == "opacity" and props[-1] == "index"
    assert header[1] == "format binary_little_endian 1.0" and "element vertex 123" in header
    assert len(raw) - raw.index(b"end_header\n") - len(b"end_header\n") == 123 * (62 * 4 + 4)
    back, bidx = model_io.load_ply(path)
    for a, b in zip(cloud.parameters(), back.parameters()):
        assert torch.equal(a.detach(), b)
    assert torch.equal(bidx, idx.long())


def test_compose_levels_semantics(tmp_path):
    """compose_models.py:41-80: carry the previous level forward, overwrite the rows listed by `index`."""
    g = torch.Generator().manual_seed(0)
    P = 400
    finest = syn.scene_1k(P=P, seed=1)
    paths = [str(tmp_path / "l0.ply")]
    model_io.save_ply(paths[0], finest)
    subsets, models = [], []
    for lvl, keep in ((1, 220), (2, 90), (3, 25)):
        idx = torch.randperm(P, generator=g)[:keep]
        sub = syn.scene_1k(P=keep, seed=10 + lvl)
        paths.append(str(tmp_path / f"l{lvl}.ply"))
        model_io.save_ply(paths[-1], sub, idx.int())
        subsets.append(idx); models.append(sub)
    fin, highest, shs_dcs, opac = model_io.compose_levels(paths)
    assert highest.shape == (P, 1) and shs_dcs.shape == (P, 4, 3) and opac.shape == (P, 4)
    assert torch.equal(fin._xyz, finest._xyz)
    exp_dc = torch.zeros(P, 4, 3); exp_op = torch.ones(P, 4); exp_hi = torch.zeros(P, 1)
    exp_dc[:, 0] = finest._features_dc[:, 0]; exp_op[:, 0] = torch.sigmoid(finest._opacity[:, 0])
    for i, (idx, sub) in enumerate(zip(subsets, models), start=1):
        exp_dc[:, i] = exp_dc[:, i - 1]; exp_op[:, i] = exp_op[:, i - 1]
        exp_dc[idx, i] = sub._features_dc[:, 0]; exp_op[idx, i] = torch.sigmoid(sub._opacity[:, 0]); exp_hi[idx] = float(i)
    assert torch.equal(shs_dcs, exp_dc) and torch.equal(opac, exp_op) and torch.equal(highest, exp_hi)
    # a Gaussian absent from level i keeps level i-1's values there
    absent = torch.ones(P, dtype=torch.bool); absent[subsets[0]] = False
    assert torch.equal(shs_dcs[absent, 1], shs_dcs[absent, 0])
    model_io.save_composed(str(tmp_path / "composed_4_12"), highest, shs_dcs, opac)
    h2, d2, o2 = model_io.load_composed(str(tmp_path / "composed_4_12"))
    assert torch.equal(h2, highest) and torch.equal(d2, shs_dcs) and torch.equal(o2, opac)


def test_reader_rejects_what_it_cannot_read(tmp_path):
    p = str(tmp_path / "bad.ply")
    open(p, "wb").write(b"ply\nformat binary_big_endian 1.0\nelement vertex 0\nend_header\n")
    with pytest.raises(ValueError):
        model_io.read_ply_vertices(p)
    open(p, "wb").write(b"not a ply")
    with pytest.raises(ValueError):
        model_io.read_ply_vertices(p)
    q = str(tmp_path / "l1.ply")
    model_io.save_ply(q, syn.scene_1k(P=5, seed=2))  # no index property
    with pytest.raises(ValueError):
        model_io.compose_levels([q, q])


@pytest.mark.gpu
def test_composed_model_renders_like_its_tensors(tmp_path):
    """Files -> compose_levels -> foveated render() equals rendering the same tensors directly."""
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    from fov3dgs_amd.gaussian_renderer_fov import render
    dev = "cuda:0"
    P = 900
    finest = syn.scene_1k(P=P, seed=6)
    highest, shs_dcs, opac = syn.foveation_layers(finest, seed=8)
    paths = [str(tmp_path / "l0.ply")]
    model_io.save_ply(paths[0], finest)
    logit = lambda o: torch.log(o / (1 - o))
    for i in range(1, 4):
        idx = torch.nonzero(highest[:, 0] >= i)[:, 0]
        sub = syn.GaussianCloud(finest._xyz[idx], shs_dcs[idx, i].unsqueeze(1).contiguous(), finest._features_rest[idx],
                                finest._scaling[idx], finest._rotation[idx], logit(opac[idx, i].clamp(1e-4, 1 - 1e-4)).unsqueeze(1))
        paths.append(str(tmp_path / f"l{i}.ply"))
        model_io.save_ply(paths[-1], sub, idx.int())
    fin, h2, d2, o2 = model_io.compose_levels(paths)
    assert torch.equal(h2, highest)
    cam = syn.camera_1k(160, 128).to(dev)
    bg = torch.zeros(3, device=dev)
    kw = dict(alpha=0.05, gazeArray=torch.tensor([0.4, 0.6]), blending=True)
    a = render(cam, fin.to(dev), bg, highest_levels=h2.to(dev), shs_dcs=d2.to(dev), opacities=o2.to(dev), **kw)["render"]
    b = render(cam, finest.to(dev), bg, highest_levels=highest.to(dev), shs_dcs=shs_dcs.to(dev),
               opacities=opac.clamp(1e-4, 1 - 1e-4).to(dev), **kw)["render"]
    # level-0 opacities go through logit/sigmoid once on the way to the file: float round trip only
    assert float((a - b).detach().abs().max()) < 1e-4
