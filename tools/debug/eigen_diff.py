"""Compare the OBB axes / half lengths k_bin keeps in its walk records with the oracle's eigen_vec / eigen_len."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.test_full_size_parity import S6M, CENTRE_WIN
from oracle import oracle as orc

s = S6M()
want = orc.forward("pcheck_obb", s.scene_plain, s.cam_dict(window=(0, 0, 1, 1)))
got = s.hip("pcheck_obb")  # same inputs as the oracle
geom = got["buffers"][0]
vid = s.native.VARIANT_IDS["pcheck_obb"]
P = s.xyz.shape[0]
def view(ptr, count, dtype):
    off = ptr - geom.data_ptr()
    return geom[off:off + 4 * count].view(dtype)
V = int(view(s.lib.fr_geometry_vis_count(vid, P, geom.data_ptr()), 1, torch.int32)[0])
wrec = view(s.lib.fr_geometry_walk_records(vid, P, geom.data_ptr()), 16 * V, torch.float32).view(V, 16).cpu().numpy()
ids_flags = wrec[:, 8].view(np.uint32)
ids = ids_flags & 0x3fffffff; flags = ids_flags >> 30
sel = (flags & 2) != 0
print("V", V, "boxtest entries", sel.sum())
i = ids[sel]
for name, hipv, orcv in (("cx", wrec[sel, 0], want["means2D"][i, 0]), ("cy", wrec[sel, 1], want["means2D"][i, 1]),
                         ("e1x", wrec[sel, 2], want["eigen_vec"][i, 0]), ("e1y", wrec[sel, 3], want["eigen_vec"][i, 1]),
                         ("e2x", wrec[sel, 4], want["eigen_vec"][i, 2]), ("e2y", wrec[sel, 5], want["eigen_vec"][i, 3]),
                         ("len1", wrec[sel, 6], want["eigen_len"][i, 0]), ("len2", wrec[sel, 7], want["eigen_len"][i, 1])):
    ne = hipv.view(np.uint32) != orcv.view(np.uint32)
    print(name, "differ bitwise:", int(ne.sum()), "max abs", float(np.abs(hipv - orcv).max()), "max ulp", int(np.abs(hipv.view(np.int32).astype(np.int64) - orcv.view(np.int32).astype(np.int64)).max()))
for g in (3754245, 908092, 2758714):
    k = np.nonzero(ids == g)[0]
    print(g, wrec[k, :8], want["means2D"][g], want["eigen_vec"][g], want["eigen_len"][g], want["conic"][g])
rec = view(s.lib.fr_geometry_records(vid, P, geom.data_ptr()), 12 * P, torch.float32).view(P, 12).cpu().numpy()
vis = want["radii"] > 0
for name, hipv, orcv in (("px", rec[vis, 0], want["means2D"][vis, 0]), ("conic_a", rec[vis, 2], want["conic"][vis, 0]), ("conic_b", rec[vis, 3], want["conic"][vis, 1]),
                         ("conic_c", rec[vis, 4], want["conic"][vis, 2]), ("depth", rec[vis, 9], want["depths"][vis]),
                         ("r", rec[vis, 6], want["rgb"][vis, 0])):
    ne = hipv.view(np.uint32) != orcv.view(np.uint32)
    print(name, "differ bitwise:", int(ne.sum()), "of", int(vis.sum()), "max ulp", int(np.abs(hipv.view(np.int32).astype(np.int64) - orcv.view(np.int32).astype(np.int64)).max()))
for g in (3754245, 908092, 2758714):
    print(g, rec[g, :6], want["conic"][g])
