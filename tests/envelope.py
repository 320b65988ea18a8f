"""What differs between two forward results of ONE frame (dicts with radii, ranges [T,2], point_list, color): the yardstick of the
FMA-contraction envelope (oracle f32 against oracle f32_fma) and of HIP against either flavour. TEST INFRASTRUCTURE ONLY.

The reference's binary was built with nvcc's default -fmad=true (its setup.py files set no -fmad=false: R0/setup.py:12-29), so
the literal reading of its fp32 expressions (the oracle's f32 flavour, -ffp-contract=off) is not the only arithmetic the source
admits. `differences(a, b)` counts what moves between two readings: Gaussians whose radius changes (and how many of them appear /
disappear), (tile, Gaussian) instances present in one frame only, tiles whose common instances come in another order, and image
values further apart than 1e-4."""
import numpy as np


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def instance_keys(res):
    """(tile << 32 | Gaussian) of every list entry, in list order."""
    rng = _np(res["ranges"]).astype(np.int64)
    lens = rng[:, 1] - rng[:, 0]
    tiles = np.repeat(np.arange(len(rng), dtype=np.int64), lens)
    pos = np.arange(int(lens.sum())) - np.repeat(np.cumsum(lens) - lens, lens) + np.repeat(rng[:, 0], lens)
    ids = _np(res["point_list"]).astype(np.int64)[pos]
    return (tiles << 32) | ids, tiles


def differences(a, b, image=True):
    ra, rb = _np(a["radii"]), _np(b["radii"])
    out = dict(gaussians=int(ra.size), radii_differ=int((ra != rb).sum()), visible_in_one_only=int(((ra > 0) != (rb > 0)).sum()))
    ka, ta = instance_keys(a)
    kb, tb = instance_keys(b)
    only_a = np.setdiff1d(ka, kb, assume_unique=True)
    only_b = np.setdiff1d(kb, ka, assume_unique=True)
    out.update(instances_a=int(ka.size), instances_b=int(kb.size), instances_in_one_only=int(only_a.size + only_b.size))
    # order: the common instances of every tile, in list order on both sides
    ca = ka[np.isin(ka, kb, assume_unique=True)]
    cb = kb[np.isin(kb, ka, assume_unique=True)]
    moved = ca != cb
    out.update(common_instances=int(ca.size), positions_in_another_order=int(moved.sum()),
               tiles_in_another_order=int(np.unique(ca[moved] >> 32).size))
    if image:
        d = np.abs(_np(a["color"]).astype(np.float64) - _np(b["color"]).astype(np.float64))
        out.update(values=int(d.size), values_gt_1e4=int((d > 1e-4).sum()), values_gt_1e5=int((d > 1e-5).sum()), max_abs=float(d.max()))
    return out
