"""3DGS model files for the rasterizer: the PLY layout the reference's GaussianModel
reads and writes, and the per-level artefacts its foveated renderer takes.

Reference behaviour restated here (no `plyfile`, which this environment lacks; numpy only):
  * GaussianModel.save_ply / save_ply_index      fov3dgs/scene/gaussian_model.py:356-398
    one binary little-endian `vertex` element, float32 properties in the order x y z nx ny nz f_dc_0..2
    f_rest_0..44 opacity scale_0..2 rot_0..3 (+ int32 `index` for the per-level models);
    f_dc / f_rest are stored channel-major (features.transpose(1, 2).flatten(1)).
  * GaussianModel.load_ply / load_ply_index      fov3dgs/scene/gaussian_model.py:454-540
    properties are looked up BY NAME (so extra / reordered properties are fine), f_rest_* and scale_* / rot*
    are ordered by their numeric suffix, f_rest is reshaped (P, 3, 15) then transposed to (P, 15, 3).
  * compose()                                     fov3dgs/compose_models.py:41-80
    level 0 = the finest model; level i copies level i-1's DC colour / opacity and overwrites the rows listed in
    level i's `index` property with that model's DC colour / sigmoid(opacity); highest_levels[index] = i.
    Results: highest_levels f32[P,1], shs_dcs f32[P,L,3], opacities f32[P,L] (what render() of
    gaussian_renderer_fov takes; the reference stores them as highest_levels.pt / shs_dcs.pt / opacities.pt).

Parity note: the reference functions need `plyfile` and its CUDA extensions to import, neither of which exists here,
so this row is checked against files written byte-for-byte in the documented layout (tests/test_model_io.py), not
against the reference's own code.
"""
import os

import numpy as np
import torch

from .synthetic import GaussianCloud

_PLY_TYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "int": "<i4", "int32": "<i4",
              "uint": "<u4", "uint32": "<u4", "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2",
              "char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1"}


def read_ply_vertices(path):
    """-> numpy structured array of the `vertex` element of a binary_little_endian (or ascii) PLY file."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt = None
        elements = []  # (name, count, [(prop, dtype)])
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: unterminated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append((tok[1], int(tok[2]), []))
            elif tok[0] == "property":
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties are not supported (element {elements[-1][0]})")
                if tok[1] not in _PLY_TYPES:
                    raise ValueError(f"{path}: unknown property type {tok[1]}")
                elements[-1][2].append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt not in ("binary_little_endian", "ascii"):
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
        out = None
        for name, count, props in elements:
            dt = np.dtype(props)
            if fmt == "ascii":
                rows = [f.readline().split() for _ in range(count)]
                arr = np.zeros(count, dtype=dt)
                for j, (pn, _) in enumerate(props):
                    arr[pn] = [r[j] for r in rows]
            else:
                arr = np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)
            if name == "vertex":
                out = arr
        if out is None:
            raise ValueError(f"{path}: no vertex element")
        return out


def _numbered(names, prefix):
    sel = [n for n in names if n.startswith(prefix)]
    return sorted(sel, key=lambda n: int(n.split("_")[-1]))


def load_ply(path, sh_degree=3, device="cpu"):
    """GaussianModel.load_ply / load_ply_index: -> (GaussianCloud, indexes int64[P] or None)."""
    v = read_ply_vertices(path)
    names = v.dtype.names
    P = v.shape[0]
    xyz = np.stack([v["x"], v["y"], v["z"]], axis=1).astype(np.float32)
    opacity = np.asarray(v["opacity"], np.float32)[:, None]
    f_dc = np.stack([v["f_dc_0"], v["f_dc_1"], v["f_dc_2"]], axis=1).astype(np.float32)[:, None, :]  # (P,1,3)
    rest_names = _numbered(names, "f_rest_")
    n_rest = (sh_degree + 1) ** 2 - 1
    if len(rest_names) != 3 * n_rest:
        raise ValueError(f"{path}: {len(rest_names)} f_rest_* properties, expected {3 * n_rest} for SH degree {sh_degree}")
    rest = np.stack([v[n] for n in rest_names], axis=1).astype(np.float32) if rest_names else np.zeros((P, 0), np.float32)
    f_rest = rest.reshape(P, 3, n_rest).transpose(0, 2, 1).copy()  # (P,15,3)
    scales = np.stack([v[n] for n in _numbered(names, "scale_")], axis=1).astype(np.float32)
    rots = np.stack([v[n] for n in _numbered(names, "rot")], axis=1).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    cloud = GaussianCloud(t(xyz), t(f_dc), t(f_rest), t(scales), t(rots), t(opacity), sh_degree=sh_degree)
    indexes = torch.from_numpy(np.asarray(v["index"], np.int64).copy()).to(device) if "index" in names else None
    return cloud, indexes


def save_ply(path, cloud, indexes=None):
    """GaussianModel.save_ply (indexes None) / save_ply_index: the layout load_ply reads."""
    xyz = cloud._xyz.detach().cpu().numpy().astype(np.float32)
    P = xyz.shape[0]
    f_dc = cloud._features_dc.detach().cpu().transpose(1, 2).flatten(start_dim=1).numpy().astype(np.float32)
    f_rest = cloud._features_rest.detach().cpu().transpose(1, 2).flatten(start_dim=1).numpy().astype(np.float32)
    cols = [("x", xyz[:, 0]), ("y", xyz[:, 1]), ("z", xyz[:, 2])] + [(n, np.zeros(P, np.float32)) for n in ("nx", "ny", "nz")]
    cols += [(f"f_dc_{i}", f_dc[:, i]) for i in range(f_dc.shape[1])]
    cols += [(f"f_rest_{i}", f_rest[:, i]) for i in range(f_rest.shape[1])]
    cols += [("opacity", cloud._opacity.detach().cpu().numpy().astype(np.float32)[:, 0])]
    sc = cloud._scaling.detach().cpu().numpy().astype(np.float32)
    ro = cloud._rotation.detach().cpu().numpy().astype(np.float32)
    cols += [(f"scale_{i}", sc[:, i]) for i in range(sc.shape[1])] + [(f"rot_{i}", ro[:, i]) for i in range(ro.shape[1])]
    dt = [(n, "<f4") for n, _ in cols]
    if indexes is not None:
        dt.append(("index", "<i4"))
    arr = np.zeros(P, dtype=dt)
    for n, c in cols:
        arr[n] = c
    if indexes is not None:
        arr["index"] = np.asarray(indexes.detach().cpu() if torch.is_tensor(indexes) else indexes, np.int32).reshape(P)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\n")
        f.write(f"element vertex {P}\n".encode())
        for n, t in dt:
            f.write(f"property {'int' if t == '<i4' else 'float'} {n}\n".encode())
        f.write(b"end_header\n")
        f.write(arr.tobytes())


def compose_levels(ply_paths, sh_degree=3):
    """compose_models.compose(): ply_paths[0] is the finest model (all P Gaussians), ply_paths[i > 0] the level-i
    models written by save_ply_index (a subset, rows addressed by `index`).
    -> (finest GaussianCloud, highest_levels f32[P,1], shs_dcs f32[P,L,3], opacities f32[P,L])"""
    L = len(ply_paths)
    finest, _ = load_ply(ply_paths[0], sh_degree)
    P = finest._xyz.shape[0]
    shs_dcs = torch.zeros((P, L, 3))
    highest_levels = torch.zeros((P, 1))
    opacities = torch.ones((P, L))
    shs_dcs[:, 0, :] = finest._features_dc[:, 0, :]
    opacities[:, 0] = torch.sigmoid(finest._opacity[:, 0])
    for i in range(1, L):
        g, idx = load_ply(ply_paths[i], sh_degree)
        if idx is None:
            raise ValueError(f"{ply_paths[i]}: level models need the `index` property (save_ply_index)")
        if idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= P):
            raise ValueError(f"{ply_paths[i]}: index out of range for a finest model of {P} Gaussians")
        shs_dcs[:, i, :] = shs_dcs[:, i - 1, :]
        shs_dcs[idx, i, :] = g._features_dc[:, 0, :]
        opacities[:, i] = opacities[:, i - 1]
        opacities[idx, i] = torch.sigmoid(g._opacity[:, 0])
        highest_levels[idx] = float(i)
    return finest, highest_levels, shs_dcs, opacities


def save_composed(folder, highest_levels, shs_dcs, opacities):
    """The three tensors as the reference stores them (compose_models.py:77-80)."""
    os.makedirs(folder, exist_ok=True)
    torch.save(highest_levels, os.path.join(folder, "highest_levels.pt"))
    torch.save(shs_dcs, os.path.join(folder, "shs_dcs.pt"))
    torch.save(opacities, os.path.join(folder, "opacities.pt"))


def load_composed(folder, device="cpu"):
    """-> (highest_levels, shs_dcs, opacities) as render_compose_gazes_fps.py:85-90 loads them."""
    return tuple(torch.load(os.path.join(folder, n), map_location=device) for n in
                 ("highest_levels.pt", "shs_dcs.pt", "opacities.pt"))
