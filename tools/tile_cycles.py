"""Per-wave timing of the foveated blend kernel (developer build: make -C fov-3dgs_amd/csrc EXTRA=-DFR_TILE_TIMERS).
Prints the distribution of wave durations, the slowest waves and how many waves are in flight over the kernel's span."""
import math, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fov3dgs_amd
from fov3dgs_amd import _native, rasterizer as rz, synthetic as syn
dev = torch.device("cuda", 0)
cloud = syn.scene_bicycle_scale(P=6_000_000, seed=1)
fov = [t.to(dev) for t in syn.foveation_layers(cloud, seed=2)]
cloud = cloud.to(dev); cam = syn.camera_ring(0, 8).to(dev)
W, H = cam.image_width, cam.image_height
with torch.no_grad():
    xyz, sc, rot = cloud.get_xyz, cloud.get_scaling.contiguous(), cloud.get_rotation.contiguous()
    rest = cloud.get_rest_features.contiguous()
rs = rz.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, device=dev),
                                      1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
E = torch.Tensor([]); vid = 3; lib = _native.load()
frames = [int(x) for x in sys.argv[1:]] or [20]
for fr in frames:
    T = ((W + 15) // 16) * ((H + 15) // 16)
    for i in range(3):
        r = rz._forward_native(vid, rs, xyz, rest, E, fov[2], sc, rot, E, fov[1], fov[0], syn.lissajous_gaze(fr, 90), 0.05, persistent=True)
        torch.cuda.synchronize()
        if i < 2:  # clear the timer records (final_T / n_contrib are not used by this variant) before the run that is read
            img = r[5]
            for ptr in (lib.fr_image_final_T(vid, W, H, img.data_ptr()), lib.fr_image_n_contrib(vid, W, H, img.data_ptr())):
                off = ptr - img.data_ptr(); img[off:off + 4 * W * H].zero_()
            torch.cuda.synchronize()
    img = r[5]
    G = 4 * T
    def view(ptr, n, dt):
        off = ptr - img.data_ptr(); return img[off:off + 4 * n].view(dt)
    ft = view(lib.fr_image_final_T(vid, W, H, img.data_ptr()), 2 * G, torch.float32).cpu().numpy()
    nc = view(lib.fr_image_n_contrib(vid, W, H, img.data_ptr()), 6 * G, torch.int32).cpu().numpy().reshape(6, G)
    dur, start = ft[:G] * 10.0, ft[G:]
    proc, batches, n, info, tloop, tsync = nc[0], nc[1], nc[2], nc[3], nc[4] * 10.0, nc[5] * 10.0
    live = (dur > 0) & (batches > 0)
    ref = start[np.nonzero(live)[0][0]]  # the first dispatched wave that ran
    st = np.where(live, ((start - ref + 1000) % (1 << 24) - 1000) * 10.0, 0.0)
    st -= st[live].min()
    end = np.where(live, st + dur, 0.0)
    dur = np.where(live, dur, 0.0)
    two = (info >> 21) & 1
    print(f"frame {fr}: waves {G}, that ran the loop {int((batches > 0).sum())}; entries in lists x waves {int(n.sum())}, staged batches {int(batches.sum())} (= {int(batches.sum()) * 64} entries), blended wave-entries {int(proc.sum())}")
    print("wave ns: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f; kernel span %.0f ns; sum of wave time / 8192 slots = %.0f ns" % (
        dur[live].mean(), *np.percentile(dur[live], [50, 90, 99, 100]), end[live].max(), dur.sum() / 8192))
    for t in np.arange(0, end[live].max(), 10000.0):
        m = (st <= t) & (end > t) & live
        print("  t=%6.0f ns: waves in flight %5d (two-level %5d) started so far %5d" % (t, int(m.sum()), int((m & (two == 1)).sum()), int(((st <= t) & live).sum())))
    for b in np.argsort(-end)[:12]:
        print("  late wave %5d tile %4d band %d lev %d two_level %d: list %4d batches %3d blended %4d start %6.0f dur %6.0f ns/blended %.0f loop %.0f topsync %.0f" % (
            b, info[b] & 0xffff, (info[b] >> 16) & 15, (info[b] >> 20) & 1, two[b], n[b], batches[b], proc[b], st[b], dur[b], dur[b] / max(proc[b], 1), tloop[b], tsync[b]))
    slow = np.argsort(-dur)[:200]
    print("  slowest 200 waves: two-level %d, list length p10 %d p50 %d max %d; rank of their tiles in the list-length order: p50 %d max %d" % (
        int(two[slow].sum()), *np.percentile(n[slow], [10, 50, 100]).astype(int),
        *np.percentile(np.searchsorted(np.sort(-n[live]), -n[slow]), [50, 100]).astype(int)))
    print("  corr(list length, dur) %.2f corr(blended, dur) %.2f" % (np.corrcoef(n[live], dur[live])[0, 1], np.corrcoef(proc[live], dur[live])[0, 1]))
    print("  blended per wave: p50 %d p90 %d p99 %d max %d; two-level waves: %d, their share of wave time %.2f" % (
        *np.percentile(proc[live], [50, 90, 99, 100]).astype(int), int((two[live] == 1).sum()), dur[two == 1].sum() / dur.sum()))
    # how far into its tile's list did a wave get (staged batches x 64)? -> what a lazy sort would have to have ready
    tile_id = info & 0xffff
    reach = np.minimum(batches.astype(np.int64) * 64, n)
    per_tile_reach = np.zeros(T, np.int64); np.maximum.at(per_tile_reach, tile_id[live], reach[live])
    per_tile_n = np.zeros(T, np.int64); np.maximum.at(per_tile_n, tile_id[live], n[live])
    for lo, hi in ((1, 513), (513, 2048), (2048, 4096), (4096, 1 << 30)):
        m = (per_tile_n >= lo) & (per_tile_n < hi)
        if m.any():
            r = per_tile_reach[m]
            print("  lists [%d,%d): %5d tiles, %8d entries, reached %8d (%.3f); reach p50 %d p90 %d p99 %d max %d; tiles reaching > 960: %d, > 1920: %d" % (
                lo, hi, int(m.sum()), int(per_tile_n[m].sum()), int(r.sum()), r.sum() / per_tile_n[m].sum(), *np.percentile(r, [50, 90, 99, 100]).astype(int),
                int((r > 960).sum()), int((r > 1920).sum())))
