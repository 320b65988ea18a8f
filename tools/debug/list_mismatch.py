"""Find (Gaussian, tile) pairs on which the HIP lists and the oracle's differ at full size; dump the oracle's per-Gaussian
state of the offenders to gpurun_out/mismatch.npz."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.test_full_size_parity import S6M, CENTRE_WIN, gaze_window, window_tiles, GX
from tests.helpers import syn
from oracle import oracle as orc

s = S6M()
rows = []
for variant, gaze, win in (("pcheck_obb", (0.5, 0.5), CENTRE_WIN), ("fov_pcheck_obb", (0.5, 0.5), gaze_window((0.5, 0.5))),
                           ("fov_pcheck_obb", syn.lissajous_gaze(47, 90), gaze_window(syn.lissajous_gaze(47, 90)))):
    scene = s.scene_fov if variant == "fov_pcheck_obb" else s.scene_plain
    want = orc.forward(variant, scene, s.cam_dict(gaze=gaze, window=win))
    got = s.hip(variant, gaze=gaze)
    g_rng = got["ranges"].cpu().numpy(); w_rng = want["ranges"].astype(np.int64)
    pl = got["point_list"].cpu().numpy()
    for t in window_tiles(win):
        a = pl[g_rng[t, 0]:g_rng[t, 1]]; b = want["point_list"][w_rng[t, 0]:w_rng[t, 1]]
        if len(a) != len(b) or (a != b).any():
            only_hip = np.setdiff1d(a, b); only_orc = np.setdiff1d(b, a)
            print(variant, gaze, "tile", t, t % GX, t // GX, "len", len(a), len(b), "only_hip", only_hip, "only_orc", only_orc)
            for g, who in [(x, 0) for x in only_hip] + [(x, 1) for x in only_orc]:
                rows.append(dict(variant=variant, tile=int(t), g=int(g), who=who, means2D=want["means2D"][g], radii=want["radii"][g],
                                 eigen_len=want["eigen_len"][g], eigen_vec=want["eigen_vec"][g], conic=want["conic"][g], depth=want["depths"][g],
                                 tile_min=want["tile_min"][t] if variant == "fov_pcheck_obb" else 0.0,
                                 hl=scene["highest_levels"][g] if variant == "fov_pcheck_obb" else 0.0,
                                 xyz=scene["means3D"][g], scale=scene["scales"][g], rot=scene["rotations"][g]))
                print(rows[-1])
np.savez(os.path.join(ROOT, "gpurun_out", "mismatch.npz"), rows=np.array(rows, dtype=object))
