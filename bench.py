#!/usr/bin/env python3
"""bench.py -- frames/sec of the foveated 1080p render on a synthetic bicycle-scale cloud.

Contract: `python bench.py --gpus N --steps K --warmup W`; rank 0 prints ONE JSON line.
  N = 1: runs in this process.
  N > 1 with the torchrun environment (RANK / WORLD_SIZE set): this process is one rank (one GPU).
  N > 1 without it: this process only LAUNCHES -- before anything here touches a GPU it starts
          `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same args>`
          as a child, relays rank 0's JSON line and exits with the child's code.

A step = one foveated frame through gaussian_renderer_fov.render() (4 layers, alpha 0.05) of the seeded S-6M cloud
 at 1920x1080 with the model handed over as the reference's tensors. Gaze of timed step i = GAZES[i % 9],
the reference's fixed set (fov3dgs/render_compose_gazes_fps.py:26: centre + 8 off-centre), so the workload does not
depend on --steps / --warmup. With N ranks each rank renders its own camera of an 8-camera ring (weak scaling; the views
are independent: no data-path collective unless --gather).

The K timed steps are run `--repeats` (5) times, each repeat bracketed by barrier + synchronize: value / ms_per_step are the
MEDIAN repeat, value_spread = [slowest, fastest]. Inside the timed frames only the HIP events around the slowest stage (found
in an untimed pass first) and around the blend stage are recorded; the other stages' times come from an untimed pass of the
same frames (an event record is a command on the stream: all eight per frame cost 3.5 %).
Besides `value` the line carries
  value_packed  the same frames with render(packed="auto") (static-model layout, bit-identical image; fovraster.h)
  value_pipelined  the same frames, the reference's tensors, TWO frames in flight (render_begin / finish on two streams: the head
                of frame n + 1 is enqueued before frame n's instance count is waited for; images bit-identical). Throughput
                mode: `value` and the reference-protocol figures stay one frame at a time, as the reference times them.
  roofline      dominant kernel: algorithmic bytes / its HIP-event duration vs the 8 TB/s HBM peak; `blend` =
                the same for the blend kernel (+ VALU / occupancy figures of the committed SQ-counter pass);
                `frame` = all stages' bytes / ms_per_step; `traffic` = PMC bytes of the committed profile if it was made
                with this very library build (traffic_source says which)
  cpu_baseline  the CPU oracle (C port of the reference algorithm, OpenMP over Gaussians / tiles) on all host cores
  stages_ms     mean per-stage kernel time of the timed frames
  extra         reference-protocol fps per gaze (events around the rasterizer only), moving-gaze fps, non-foveated
                forward fps, training step fwd / bwd / loss ms (median of 50); N > 1: frames/s with every frame gathered on
                rank 0 and a multi-view training pass with its collective time (the headline's views are independent)
--mode train (config 5): every rank does forward + loss + backward of its camera (pcheck_obb_sum, fused L1+SSIM) and the
gradients are summed over ranks; reports fwd_bwd_ms and collective_ms.
"""
import argparse
import gc
import hashlib
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
GAZES = [(0.25 * i, 0.25 * j) for i in range(1, 4) for j in range(1, 4)]  # render_compose_gazes_fps.py:26
PROFILE_TAG = "r06"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=63)
    ap.add_argument("--warmup", type=int, default=9)
    ap.add_argument("--mode", choices=("render", "train"), default="render")
    ap.add_argument("--points", type=int, default=6_000_000)
    ap.add_argument("--cloud", choices=("S-6M", "S-6M-T"), default="S-6M",
                    help="S-6M: the headline cloud (SURVEY 8d). S-6M-T: the same cloud with opacities that make the blend consume its lists "
                         "(synthetic.scene_translucent) as the main workload -- profiling passes (tools/make_profiles.sh); the default run "
                         "reports S-6M-T under extra.translucent")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--repeats", type=int, default=5,
                    help="the K timed steps are run this many times (each bracketed by barrier + synchronize); value = the median "
                         "repeat, value_spread = [min, max] (one 16 ms region cannot rank two builds)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--packed-only", action="store_true", help="time only the packed-model frames (profiling passes)")
    ap.add_argument("--headline-only", action="store_true",
                    help="time only the headline frames (the reference's tensors, one frame at a time): no packed-layout and no "
                         "two-frames-in-flight runs (profiling passes: one workload per kernel-stats file)")
    ap.add_argument("--serial-only", action="store_true",
                    help="one frame on the GPU at a time for the whole run (rasterizer.OVERLAP_SUCCESSIVE_FRAMES off): value == value_serial. "
                         "The profiling passes: a kernel's duration and counters are then its own (tools/make_profiles.sh)")
    ap.add_argument("--gather", action="store_true",
                    help="N > 1: also collect every rank's frame on rank 0 (asynchronous RCCL gather overlapped with the next "
                         "frame). Off by default: the views are independent and the path has no exchange step.")
    ap.add_argument("--activation-pass", action="store_true",
                    help="--mode train: activate the model's parameters with the fused pass instead of inside the rasterizer's kernels")
    ap.add_argument("--row-sparse", action="store_true",
                    help="--mode train: row-sparse gradients (an extension; the reference's contract, and the default, is dense)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launch / rendezvous only (gloo without a GPU): rank 0 prints a line with n_gpus and exits")
    ap.add_argument("--master-port", type=int, default=0)
    return ap.parse_args(argv)


class Timed(tuple):
    """(seconds of K frames with the nine gazes weighted equally -- the median repeat, [min, max] over the repeats, {stage: mean ms})
    + .raw: the same repeat's wall clock between the two barrier + synchronize brackets (gaze i % 9: K % 9 gazes weigh one frame more)."""

    def __new__(cls, seconds, spread, stages, raw):
        self = super().__new__(cls, (seconds, spread, stages))
        self.raw = raw
        return self


def launch_ranks(args):
    """--gpus N > 1 outside torchrun: start the N ranks as a child process tree (this process never touches a GPU)."""
    import socket
    port = args.master_port
    if not port:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout:
        if ln.startswith('{"metric"'):
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: the ranks exited without printing a result line\n")
        rc = 1
    sys.exit(rc)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class Pipe:
    debug = False


def lib_sha16():
    from fov3dgs_amd import _native
    with open(_native.LIB_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def algorithmic_bytes(variant, n):
    """SURVEY.md 8(d) per-unit figures x the units of one launch, per stage of this build, for the FOVEATED (RF) or a plain
    (R0 / RS / RP) frame. n: P, Px, T, V_in (in front of the near plane), C (candidates: survivors of the conservative cull
    pass), V (visible: radii > 0), D (instances), D_single / D_blend (instances in single- / two-level tiles).
    RF (RF/cuda_rasterizer/forward.cu:105-238, rasterizer_impl.cu:264-383, 490-530) evaluates SH only for the Gaussians that
    SURVIVE the filter and per level in their range; R0 / RS / RP (forward.cu:155-262) for every Gaussian inside the frustum:
      project   12 B xyz read + 8 B radii / tiles_touched written per Gaussian; scale 12 + rotation 16 (+ RF highest level 4)
                read per Gaussian in front of the camera
      bin       (the reference's preprocess / filter / compute_fov_colors; this build: k_bin = full projection of the candidates, tile
                counts, colours)
                RF: the candidate's inputs handed on (48 B per candidate) and, per VISIBLE Gaussian, depth 4 + mean2D 8 + conic 12
                + eigen axes 24 + walk record 64 written; 180 (rest SH) + 48 (level DCs) + 16 (level opacities) + 4 (highest level)
                read, level range 8 + four level rows 64 written: 48 C + 432 V in all
                plain: 48 (+ 24 eigen) written per visible Gaussian; opacity 4 + SH 192 read per Gaussian in front of the
                camera (B_pre's 224 V_in less the 28 of project)
      emit      12 B per instance (key + value) + 44 B read per visible Gaussian;  tile_sort: 12 B per instance (this build moves
                (depth, item) once per stage instead of a 6-pass radix sort);  tile_scan 16 T;  tile_levels 20 T
      render    RF: 32 D_single + 52 D_blend + 12 Px;  plain: 40 D + 20 Px"""
    fov = variant == "fov_pcheck_obb"
    b = {"project": 20 * n["P"] + (32 if fov else 28) * n["V_in"],
         "emit": 12 * n["D"] + 44 * n["V"], "tile_sort": 12 * n["D"], "tile_scan": 16 * n["T"], "tile_levels": 20 * n["T"] if fov else 0}
    if fov:
        b["bin"] = 48 * n["C"] + (112 + 248 + 72) * n["V"]
        b["render"] = 32 * n["D_single"] + 52 * n["D_blend"] + 12 * n["Px"]
    else:
        b["bin"] = (48 + 24) * n["V"] + 196 * n["V_in"]
        b["render"] = 40 * n["D"] + 20 * n["Px"]
    return b


def training_bytes(n, sh_coeffs=16):
    """SURVEY.md 8(d) for the kernels of a training step beside the forward stages above (n as in algorithmic_bytes, plain frame):
      render (RS)      40 D + 20 Px        R0/RS forward blend (RS/cuda_rasterizer/forward.cu:298-430)
      render_bwd       40 D + 20 Px + 80 V (R0/cuda_rasterizer/backward.cu:399-557)
      preprocess_bwd   276 V + 256 V       (backward.cu:144-396: inputs re-read, gradients written)
      fill_zero        what RasterizeGaussiansBackwardCUDA's torch::zeros clear of the tensors this call returns
                       (RS/rasterize_points.cu:171-179): dL_dmeans3D 12 + dL_dmeans2D 12 + dL_dopacity 4 + dL_dscales 12 +
                       dL_drotations 16 + dL_dsh 12 M bytes per Gaussian
    The same bytes split the way the library's kernels write them (csrc/backward.hip SmallSet; split SH storage, as the step here
    uses): the zeros of the five narrow tensors and of the DC coefficients (56 + 12 = 68 bytes per Gaussian without a gradient row)
    leave with the rows, from k_preprocess_bwd; k_fill_zero clears the rest coefficients' tensor (12 (M - 1) bytes per Gaussian)."""
    narrow = 56 + 12
    return {"render": 40 * n["D"] + 20 * n["Px"], "render_bwd": 40 * n["D"] + 20 * n["Px"] + 80 * n["V"],
            "preprocess_bwd": 532 * n["V"] + narrow * (n["P"] - n["V"]), "fill_zero": 12 * (sh_coeffs - 1) * n["P"]}


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        launch_ranks(args)  # does not return
    import numpy as np
    import torch
    import fov3dgs_amd  # noqa: F401
    from fov3dgs_amd import _native, multiview, synthetic as syn
    from fov3dgs_amd.gaussian_renderer import render as render_plain
    from fov3dgs_amd.gaussian_renderer_fov import render as render_fov
    from fov3dgs_amd.profiling import StageTimer

    rank, world, local_rank = multiview.init_distributed()
    if args.dry_launch:
        if world > 1:
            t = torch.ones(1)
            torch.distributed.all_reduce(t)  # the rendezvous works
            assert int(t.item()) == world
        if rank == 0:
            print(json.dumps({"metric": "frames/sec at 1080p foveated (bicycle-scale)", "value": None, "unit": "frames/s",
                              "n_gpus": world, "dry_launch": True, "backend": torch.distributed.get_backend() if world > 1 else None}),
                  flush=True)
        return
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    lib = _native.load()
    K, Wm = args.steps, args.warmup
    W, H = args.width, args.height
    gx, gy = (W + 15) // 16, (H + 15) // 16
    T = gx * gy

    def barrier_sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    t0 = time.time()

    class FrozenCloud:
        """Inference-time view of a cloud: activations evaluated once, getters return resident tensors."""

        def __init__(self, c):
            with torch.no_grad():
                self.get_xyz = c.get_xyz.detach()
                self.get_scaling = c.get_scaling.detach().contiguous()
                self.get_rotation = c.get_rotation.detach().contiguous()
                self.get_opacity = c.get_opacity.detach().contiguous()
                self.get_features = c.get_features.detach().contiguous()
                self.get_rest_features = c.get_rest_features.detach().contiguous()
            self.get_features_detach_rest = self.get_features
            self.active_sh_degree = c.active_sh_degree

    def make_scene(opacity_logit):
        """-> dict: the seeded cloud (CPU + device), its foveation layers and the inference-time view of it"""
        c_cpu = syn.scene_bicycle_scale(P=args.points, seed=1, opacity_logit=opacity_logit)
        f_cpu = syn.foveation_layers(c_cpu, seed=2)
        c_dev = c_cpu.to(dev)
        hl, dcs, op4 = [t.to(dev) for t in f_cpu]
        return dict(cloud_cpu=c_cpu, fov_cpu=f_cpu, cloud=c_dev, pc=FrozenCloud(c_dev), highest=hl, shs_dcs=dcs, opac=op4)

    scene = make_scene(syn.OPACITY_LOGIT_S6MT if args.cloud == "S-6M-T" else syn.OPACITY_LOGIT_S6M)
    cloud_cpu, fov_cpu, cloud, pc = scene["cloud_cpu"], scene["fov_cpu"], scene["cloud"], scene["pc"]
    highest, shs_dcs, opac = scene["highest"], scene["shs_dcs"], scene["opac"]
    n_views = 8
    my_view = rank % n_views
    cam = syn.camera_ring(my_view, n_views, W, H).to(dev)
    bg = torch.zeros(3, device=dev)
    if rank == 0:
        log(f"[bench] scene ready in {time.time() - t0:.1f}s: P={args.points} {W}x{H} world={world} mode={args.mode}")

    if args.mode == "train":
        return train_mode(args, rank, world, dev, cloud, cam, bg, multiview, render_plain, barrier_sync)
    cloud_name = args.cloud

    def frame_of(sc):
        def frame_fn(gaze, packed, **kw):
            return render_fov(cam, sc["pc"], bg, alpha=0.05, gazeArray=gaze, blending=True, highest_levels=sc["highest"],
                              shs_dcs=sc["shs_dcs"], opacities=sc["opac"], packed=packed, **kw)
        return frame_fn
    frame = frame_of(scene)

    settled = set()

    def timed_run(packed, event_stages, frame=frame, repeats=None):
        """W warm-ups + `repeats` x exactly K timed frames (gaze i % 9), barrier + synchronize on both sides of every repeat.
        Only the boundaries of `event_stages` are recorded inside the timed frames (every event record is a command on the
        stream, ~3 us: all eight cost 3.5 % of the frame). -> (median seconds, [min, max] seconds, {stage: mean ms})"""
        pending = None
        times, balanced, per_stage = [], [], {k: [] for k in event_stages}
        with torch.no_grad():
            # W warm-up frames -- and, the first time a mode runs, enough of them (45) for the runtime's one-time work to be over:
            # the internal streams' queues, the caching allocator's pool of output blocks (a 36-ms hiccup within the first forty
            # frames of a process otherwise lands in the first timed repeat)
            settle = max(Wm, 45) if (packed, id(frame)) not in settled else Wm
            settled.add((packed, id(frame)))
            for i in range(settle):
                out = frame(GAZES[i % 9], packed)
                if world > 1 and args.gather:
                    multiview.gather_images(out["render"], dst=0)
            for rep in range(max(1, args.repeats if repeats is None else repeats)):
                barrier_sync()
                timer = StageTimer(K, stages=event_stages)
                slots = np.zeros(K)
                gc_was = gc.isenabled()
                if os.environ.get("FOVRASTER_BENCH_GC") != "1":
                    gc.disable()  # (as timeit does: a collection of this process's many objects in the middle of a frame is not the rasterizer's time)
                t_start = t_prev = time.perf_counter()
                with timer:
                    for i in range(K):
                        out = frame(GAZES[i % 9], packed)
                        if world > 1 and args.gather:
                            if pending is not None:
                                pending[0].wait()
                            pending = multiview.gather_images(out["render"], dst=0, async_op=True) + (out["render"],)
                        t_now = time.perf_counter()
                        slots[i], t_prev = t_now - t_prev, t_now
                    if pending is not None:
                        pending[0].wait()
                        pending = None
                barrier_sync()
                t_end = time.perf_counter()
                if gc_was:
                    gc.enable()
                elapsed = t_end - t_start
                # a call returns once its instance count is in: slot i holds frame i's head and frame i - 1's tail. The last frame's
                # tail (drained by the synchronize above) goes to slot 0, which had no tail in front of it: sum(slots) == elapsed.
                slots[0] += t_end - t_prev
                per_gaze = [float(np.mean(slots[g::9])) for g in range(min(9, K))]
                bal = float(np.mean(per_gaze)) * K   # seconds K frames take when every gaze has the same weight
                if world > 1:
                    t = torch.tensor([elapsed, bal], device=dev, dtype=torch.float64)
                    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                    elapsed, bal = float(t[0].item()), float(t[1].item())
                times.append(elapsed)
                balanced.append(bal)
                if os.environ.get("FOVRASTER_BENCH_DEBUG"):
                    log(f"[bench] repeat {rep}: {elapsed * 1e3:.3f} ms raw, {bal * 1e3:.3f} balanced, slowest frames {np.sort(slots)[-3:] * 1e3}")
                for row in timer.stage_ms():
                    for k in event_stages:
                        per_stage[k].append(row[k])
                timer.close()
        mid = int(np.argsort(balanced)[len(balanced) // 2])
        return Timed(balanced[mid], [min(balanced), max(balanced)], {k: float(np.mean(v)) for k, v in per_stage.items()}, times[mid])

    def timed_run_pipelined(depth=2):
        """The same K frames per repeat with `depth` frames in flight (render_begin / finish, one stream per slot)."""
        from fov3dgs_amd.gaussian_renderer_fov import render_begin
        streams = [torch.cuda.Stream(dev) for _ in range(depth)]
        kw = dict(alpha=0.05, blending=True, highest_levels=highest, shs_dcs=shs_dcs, opacities=opac)

        def run(n):
            pending = []
            for i in range(n):
                pending.append(render_begin(cam, pc, bg, gazeArray=GAZES[i % 9], stream=streams[i % depth], **kw))
                if len(pending) == depth:
                    pending.pop(0).finish()
            for p_ in pending:
                p_.finish()
        torch.cuda.synchronize()
        run(Wm)
        times = []
        for rep in range(max(1, args.repeats)):
            barrier_sync()
            t_start = time.perf_counter()
            run(K)
            barrier_sync()
            el = time.perf_counter() - t_start
            if world > 1:
                t = torch.tensor([el], device=dev, dtype=torch.float64)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                el = float(t.item())
            times.append(el)
        return float(np.median(times)), [min(times), max(times)]

    def stage_pass(packed, n=27, frame=frame):
        """Untimed: every stage's kernel time (all eight events per frame), mean over n frames of the nine gazes."""
        with torch.no_grad():
            timer = StageTimer(n)
            with timer:
                for i in range(n):
                    frame(GAZES[i % 9], packed)
            torch.cuda.synchronize()
            st = timer.stage_ms()
            timer.close()
        return {k: float(np.mean([s_[k] for s_ in st])) for k in _native.STAGES}

    # which stage is the slowest is found first (untimed); the timed frames then carry the events of that stage and of the
    # blend stage (the kernel north_star names) only
    from fov3dgs_amd import rasterizer as rz_mod
    if args.serial_only:
        rz_mod.OVERLAP_SUCCESSIVE_FRAMES = False
    overlap_default = bool(rz_mod.OVERLAP_SUCCESSIVE_FRAMES)
    _stage_pass = stage_pass

    def stage_pass(*a_, **k_):
        """(per-stage kernel times are a frame's own: one frame on the GPU at a time)"""
        rz_mod.OVERLAP_SUCCESSIVE_FRAMES = False
        try:
            return _stage_pass(*a_, **k_)
        finally:
            rz_mod.OVERLAP_SUCCESSIVE_FRAMES = overlap_default
    with torch.no_grad():
        for i in range(3):
            frame(GAZES[i % 9], "auto" if args.packed_only else None)
    pre = stage_pass("auto" if args.packed_only else None)
    dominant = max(("project", "bin", "render", "tile_sort", "emit"), key=lambda k: pre[k])
    ev_stages = tuple(dict.fromkeys((dominant, "render")))
    if args.packed_only:
        headline = serial = timed_run("auto", ev_stages)
        elapsed_p, spread_p, timed_ms_p = headline
        elapsed, spread, timed_ms, elapsed_raw = elapsed_p, spread_p, timed_ms_p, headline.raw
        timed_ms_overlapped = timed_ms
    else:
        # one frame on the GPU at a time (rasterizer.OVERLAP_SUCCESSIVE_FRAMES off): a kernel's duration is its own -- the roofline's
        # timed region -- and `value_serial`
        rz_mod.OVERLAP_SUCCESSIVE_FRAMES = False
        serial = timed_run(None, ev_stages)
        rz_mod.OVERLAP_SUCCESSIVE_FRAMES = overlap_default
        # render() as a caller gets it: the headline. No stage events inside these frames: an event record is a barrier packet with a
        # cache write-back on its stream, and three frames share the GPU here (with the two stages' four records per frame the same
        # frames ran 8 % slower)
        headline = timed_run(None, ()) if overlap_default else serial
        elapsed, spread, timed_ms_overlapped = headline
        timed_ms = serial[2]
        elapsed_raw = headline.raw
        if args.headline_only:
            elapsed_p, spread_p, timed_ms_p = elapsed, spread, timed_ms
        else:
            elapsed_p, spread_p, timed_ms_p = timed_run("auto", ev_stages)  # static-model layout
    mean_ms = stage_pass("auto" if args.packed_only else None)
    mean_ms_p = mean_ms if args.headline_only else stage_pass("auto")
    # (last: its two launch streams and their helper pairs stay in existence, and a process with more streams than hardware queues
    # -- four by default -- shares queues between them)
    elapsed_pl, spread_pl = (elapsed, spread) if (args.headline_only or args.packed_only) else timed_run_pipelined(2)
    multi = multi_gpu_extras(args, rank, world, dev, cloud, cam, bg, multiview, render_plain, barrier_sync, frame, torch, np) if world > 1 else {}
    if rank != 0:
        return

    # ---- untimed post-pass: instance statistics of the timed gazes (for the algorithmic bytes) ----
    vid = _native.VARIANT_FOV_PCHECK_OBB
    from fov3dgs_amd import rasterizer as rz

    def gaze_stats(sc):
        """-> (mean over the timed steps' gazes, per-gaze list, V_in) of V, C, D, D_single, D_blend, max_list, and `consumed`: the
        fraction of the frame's instances the blend fetches before its pixels are finished (fr_forward_args.list_consumed)."""
        per = []
        pc_ = sc["pc"]
        cons = torch.zeros(T, dtype=torch.int32, device=dev)
        pairs = torch.zeros(T, dtype=torch.int32, device=dev)
        with torch.no_grad():
            for gaze in GAZES:
                rs = rz.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg, 1.0,
                                                      cam.world_view_transform, cam.full_proj_transform, 3,
                                                      cam.camera_center, False, False)
                res = rz._forward_native(vid, rs, pc_.get_xyz, pc_.get_rest_features, torch.Tensor([]), sc["opac"], pc_.get_scaling,
                                         pc_.get_rotation, torch.Tensor([]), sc["shs_dcs"], sc["highest"], gaze, 0.05, list_consumed=cons, blend_pairs=pairs)
                torch.cuda.synchronize()
                d = frame_stats(torch, lib, vid, (res[0], res[2], res[5]), W, H, T, geom=res[3], P=args.points)
                d["consumed"] = float(cons.sum().item()) / max(d["D"], 1)
                d["pairs"] = int(pairs.long().sum().item())  # (band of eight rows, list entry) pairs the blend evaluated
                per.append(d)
            vm = cam.world_view_transform
            z = pc_.get_xyz @ vm[:3, 2] + vm[3, 2]
            v_in = int((z > 0.2).sum().item())
        # `value` weighs the gazes the timed steps cover equally (timed_run): so do the per-gaze statistics
        wts = np.array([1.0 if g < K else 0.0 for g in range(9)], dtype=np.float64)
        mean = {k: float(np.sum([s_[k] * w for s_, w in zip(per, wts)]) / wts.sum()) for k in per[0]}
        return mean, per, v_in
    st, stats, V_in = gaze_stats(scene)
    P, Px = args.points, W * H
    counts = dict(st, P=P, Px=Px, T=T, V_in=V_in)
    alg_bytes = algorithmic_bytes("fov_pcheck_obb", counts)  # the RF formulas: SH per VISIBLE Gaussian (see there)
    prof = load_profiles()
    # kernel durations of the roofline: the events recorded INSIDE the timed frames for the two stages that carry them
    # (measured_in says so), the untimed all-stage pass of the same frames for the rest
    roof_ms = dict(mean_ms)
    roof_ms.update(timed_ms)

    def roof(stage):
        ach = alg_bytes[stage] / (roof_ms[stage] * 1e-3) / 1e9
        d = dict(kernel=stage, achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 5),
                 algorithmic_bytes=int(alg_bytes[stage]), kernel_ms=round(roof_ms[stage], 4),
                 measured_in=("timed region of value_serial (HIP events on the launch stream; one frame on the GPU at a time: the kernel's duration is its own)"
                              if stage in timed_ms else "untimed all-stage pass of the same frames"))
        if stage in timed_ms_overlapped and overlap_default:
            d["kernel_ms_overlapped"] = round(timed_ms_overlapped[stage], 4)  # the same kernel in the timed region of `value`, sharing the GPU with the neighbouring frame's tail / head
        d["traffic"], d["traffic_source"] = prof.traffic(stage, packed=args.packed_only)
        if d["traffic"]:
            # the formula charges bytes this build never moves (SH rows of culled Gaussians) or moves twice: the PMC bytes over the
            # same duration say how fast the memory system really ran
            d["frac_traffic"] = round(d["traffic"] / (roof_ms[stage] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
        return d
    roofline = dict(bound="hbm", **roof(dominant))
    roofline["blend"] = roof("render")
    roofline["blend"].update(prof.blend_sq())
    if roofline["blend"].get("valu_insts"):
        roofline["blend"]["valu_insts_per_blend_pair"] = round(roofline["blend"]["valu_insts"] / max(st["pairs"], 1), 1)
    frame_bytes = sum(alg_bytes.values())
    ms_step = elapsed / K * 1e3
    roofline["frame"] = dict(algorithmic_bytes=int(frame_bytes), ms=round(ms_step, 4),
                             achieved=round(frame_bytes / (ms_step * 1e-3) / 1e9, 2), frac=round(frame_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5))
    roofline["per_kernel"] = {k: dict(ms=round(roof_ms[k], 4), alg_GBs=round(alg_bytes[k] / max(roof_ms[k], 1e-9) / 1e6, 1))
                              for k in _native.STAGES}

    extra = {}
    if world == 1 and not args.no_extra:
        extra = extras(args, torch, np, syn, dev, cam, pc, cloud, bg, frame, render_plain, H, W)
    if world == 1:
        extra["canary"] = canary(torch, dev, mean_ms, serial[0] / K * 1e3)

    # ---- S-6M-T: the same frames and the same training step on a cloud that CONSUMES its lists (synthetic.scene_translucent: same
    # geometry, seeds and SH, opacity logits ~ N(-3.5, 1): the blend fetches 0.9 of a foveated frame's instances and 0.7 of the
    # training frame's, 1.0 M Gaussians receive a gradient) -- which stage leads depends on the workload, and a trained model is
    # nearer to this one than to S-6M's saturating cloud
    if world == 1 and not args.no_extra and args.cloud == "S-6M":
        tsc = make_scene(syn.OPACITY_LOGIT_S6MT)
        frame_t = frame_of(tsc)
        with torch.no_grad():
            for i in range(9):
                frame_t(GAZES[i], None)
        el_t, sp_t, _ = timed_run(None, (), frame=frame_t, repeats=3)
        ms_t = stage_pass(None, frame=frame_t)
        st_t, stats_t, v_in_t = gaze_stats(tsc)
        bytes_t = algorithmic_bytes("fov_pcheck_obb", dict(st_t, P=P, Px=Px, T=T, V_in=v_in_t))
        slow = max(("project", "bin", "render", "tile_sort", "emit"), key=lambda k: ms_t[k])
        tr_t = tsc["cloud"].requires_grad_(True)
        target_t = torch.rand(3, H, W, device=dev)
        tt = training_extras(torch, np, syn, render_plain, cam, bg, tr_t, target_t, H, W, short=True)
        tt.pop("train_note", None)
        # the plain (training) frame's consumed share and the rows that receive a gradient
        cons = torch.zeros(T, dtype=torch.int32, device=dev)
        tpc = tsc["pc"]
        with torch.no_grad():
            rs_p = rz.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg, 1.0, cam.world_view_transform,
                                                    cam.full_proj_transform, 3, cam.camera_center, False, False)
            r_p = rz._forward_native(_native.VARIANT_PCHECK_OBB_SUM, rs_p, tpc.get_xyz, tpc.get_features, torch.Tensor([]), tpc.get_opacity,
                                     tpc.get_scaling, tpc.get_rotation, torch.Tensor([]), persistent=True, list_consumed=cons)
            torch.cuda.synchronize()
            plain_consumed = float(cons.sum().item()) / max(int(r_p[0]), 1)
        grad_rows = int((tr_t._opacity.grad != 0).sum().item()) if tr_t._opacity.grad is not None else None
        extra["translucent"] = dict(
            workload="S-6M-T: the S-6M cloud with opacity logits ~ N(%.1f, %.1f^2) (synthetic.scene_translucent), same camera, gazes, protocol" % syn.OPACITY_LOGIT_S6MT,
            fps=round(K / el_t, 2), ms_per_step=round(el_t / K * 1e3, 4), fps_spread=[round(K / sp_t[1], 2), round(K / sp_t[0], 2)],
            stages_ms={k: round(v, 4) for k, v in ms_t.items()}, list_consumed_frac=round(st_t["consumed"], 4), blend_pairs=int(st_t["pairs"]),
            list_consumed_frac_training_frame=round(plain_consumed, 4), gaussians_with_gradient=grad_rows,
            visible=int(st_t["V"]), instances=int(st_t["D"]), max_tile_list=int(max(s_["max_list"] for s_ in stats_t)),
            roofline=dict(kernel=slow, bound="hbm", kernel_ms=round(ms_t[slow], 4), algorithmic_bytes=int(bytes_t[slow]),
                          achieved=round(bytes_t[slow] / (ms_t[slow] * 1e-3) / 1e9, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                          frac=round(bytes_t[slow] / (ms_t[slow] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                          note="the blend is VALU-bound (DESIGN 4): its HBM fraction is reported by the same formula as every stage's"),
            **tt)
        for p_ in tr_t.parameters():
            p_.grad = None
        del tsc, tr_t, tpc

    if "_train_counts" in extra:
        n_tr, fwd_ms, bwd_ms = extra.pop("_train_counts"), extra.pop("_train_fwd_ms"), extra.pop("_train_bwd_ms")
        tb = training_bytes(n_tr)
        fb = algorithmic_bytes("pcheck_obb_sum", dict(n_tr, C=0, D_single=0, D_blend=0))
        tr_ms = dict(fwd_ms, **bwd_ms)
        kern = {"render": "k_render", "render_bwd": "k_render_bwd", "preprocess_bwd": "k_preprocess_bwd", "fill_zero": "k_fill_zero",
                "project": "k_project", "bin": "k_bin", "emit": "k_emit", "tile_sort": "k_tile_msort*"}
        rows = {}
        for k_, byt in list(tb.items()) + [(k2, fb[k2]) for k2 in ("project", "bin", "emit", "tile_sort")]:
            ms_ = tr_ms[k_]
            if not (isinstance(ms_, float) and math.isfinite(ms_) and ms_ > 0):
                continue
            tr_b, tr_src = prof.traffic_train(kern[k_])
            rows[k_] = dict(kernel=kern[k_], ms=round(ms_, 4), algorithmic_bytes=int(byt), achieved=round(byt / max(ms_, 1e-9) / 1e6, 1),
                            frac=round(byt / max(ms_, 1e-9) / 1e6 / HBM_PEAK_GBS, 5), traffic=tr_b,
                            frac_traffic=None if not tr_b else round(tr_b / max(ms_, 1e-9) / 1e6 / HBM_PEAK_GBS, 5))
        roofline["train"] = dict(unit="GB/s", peak=HBM_PEAK_GBS, counts=n_tr, kernels=rows,
                                 note="training step (pcheck_obb_sum, raw parameters, fused L1 + SSIM): median of 10 instrumented steps, HIP events "
                                      "recorded by the library on the streams the kernels run on (fill_zero: the zero fill of the rest coefficients' gradient tensor + the empty "
                                      "groups of the narrow ones, on a helper stream beside render_bwd); bytes: "
                                      "training_bytes() / algorithmic_bytes() of the plain frame; traffic: " + str(prof.train_source()))
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(cloud_cpu, fov_cpu, syn.camera_ring(my_view, n_views, W, H), GAZES[4], T, gx, gy)
        except Exception as e:  # the baseline must never take the bench line down
            cpu = dict(value=None, unit="frames/s", cores=os.cpu_count(), kind="port", sample=f"failed: {e}")

    line = {
        "metric": "frames/sec at 1080p foveated (bicycle-scale)", "value": round(world * K / elapsed, 3), "unit": "frames/s",
        "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": round(ms_step, 4),
        "value_raw": round(world * K / elapsed_raw, 3), "ms_per_step_raw": round(elapsed_raw / K * 1e3, 4),
        "value_serial": round(world * K / serial[0], 3), "ms_per_step_serial": round(serial[0] / K * 1e3, 4),
        "value_serial_spread": [round(world * K / serial[1][1], 3), round(world * K / serial[1][0], 3)],
        "overlap_note": "value: render() as a caller gets it -- one call at a time, each returning its instance count; successive inference calls whose inputs "
                        "are provably unchanged (same tensor objects, addresses, autograd versions) run on three internal streams in turn, so the head of call "
                        "n + 1 (cull, projection, counts) runs beside the tail of call n (emission, sort, blend): rasterizer.OVERLAP_SUCCESSIVE_FRAMES = "
                        + str(overlap_default) + "; images bit-identical. value_serial: the same K frames with that switch off (every kernel of a frame alone on the GPU: "
                        "the roofline's kernel durations are measured there)",
        "value_note": "value / ms_per_step: the K timed frames (barrier + synchronize on both sides) with the nine gazes weighted EQUALLY -- "
                      "mean over the gazes of the mean time of that gaze's frames inside the region -- so that it does not depend on steps % 9; "
                      "value_raw / ms_per_step_raw: K / the region's wall clock (gaze i % 9: steps % 9 gazes weigh one frame more)",
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "repeats": max(1, args.repeats), "value_spread": [round(world * K / spread[1], 3), round(world * K / spread[0], 3)],
        "value_packed": round(world * K / elapsed_p, 3), "ms_per_step_packed": round(elapsed_p / K * 1e3, 4),
        "value_packed_spread": [round(world * K / spread_p[1], 3), round(world * K / spread_p[0], 3)],
        "value_pipelined": round(world * K / elapsed_pl, 3), "ms_per_step_pipelined": round(elapsed_pl / K * 1e3, 4),
        "value_pipelined_spread": [round(world * K / spread_pl[1], 3), round(world * K / spread_pl[0], 3)],
        "pipeline_depth": 2,
        "config": {"workload": cloud_name + " bicycle-scale cloud, 4-layer foveated render (fov_pcheck_obb), the reference's 9 fixed gazes "
                               "(0.25 i, 0.25 j) in turn, one camera per GPU" + (", frames gathered on rank 0" if (world > 1 and args.gather) else ""),
                   "gaussians": P, "width": W, "height": H, "alpha": 0.05, "sh_degree": 3,
                   "visible": int(st["V"]), "candidates": int(st["C"]), "in_front": V_in, "instances": int(st["D"]),
                   "instances_blend_tiles": int(st["D_blend"]), "max_tile_list": int(max(s["max_list"] for s in stats)),
                   "list_consumed_frac": round(st["consumed"], 4), "blend_pairs": int(st["pairs"]),
                   "list_consumed_note": "share of the frames' sorted instances the blend fetches before every pixel of their tile is finished "
                                         "(fr_forward_args.list_consumed, batches of 64): the S-6M cloud (SURVEY 8d: opacity = sigmoid(N(1, 2^2))) "
                                         "saturates early; extra.translucent is the same frame on S-6M-T, a cloud that consumes its lists",
                   "model_layout": "value: the reference's tensor interface (render(packed=None)); value_packed: the static-model "
                                   "layout (packed_geom / packed_colour, made once, bit-identical image)",
                   "parallelism": f"views{world}"},
        "roofline": roofline,
        "cpu_baseline": cpu,
        "stages_ms": {k: round(v, 4) for k, v in mean_ms.items()},
        "stages_ms_packed": {k: round(v, 4) for k, v in mean_ms_p.items()},
        "extra": dict(extra, **multi),
    }
    print(json.dumps(_finite(line)), flush=True)


CANARY_PROJECT_US = 75.0  # k_project on a box in the faster of the two memory states the pool shows (75 / 81 us)


def canary(torch, dev, stages_ms, ms_step):
    """What state this box's memory system is in (boxes of the pool, and one box before / after minutes of load, run the memory-bound
    stages 4-7 % apart): a 1-GiB device-to-device copy (streaming read + write, HIP events) and the cull pass's kernel time,
    the frame's own streaming kernel. `value_serial_normalised`: frames/s of the one-frame-at-a-time run had the two memory-bound stages (project, bin) run at the reference
    state's speed -- both scaled by CANARY_PROJECT_US / this run's k_project time; a yardstick for comparing lines across boxes, never `value`."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty(n, dtype=torch.uint8, device=dev)
    a.zero_()
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del a, b
    proj_us = stages_ms["project"] * 1e3
    scale = CANARY_PROJECT_US / max(proj_us, 1e-9)
    ms_norm = ms_step - (stages_ms["project"] + stages_ms["bin"]) * (1.0 - scale)
    return dict(copy_1GiB_GBs=round(2 * n / (ms * 1e-3) / 1e9, 1), copy_ms=round(ms, 4), k_project_us=round(proj_us, 2),
                reference_k_project_us=CANARY_PROJECT_US, value_serial_normalised=round(1e3 / ms_norm, 1),
                note="value_serial_normalised = 1000 / (ms_per_step_serial - (project + bin) * (1 - 75 us / k_project_us)): value_serial with the two "
                     "memory-bound stages scaled to the pool's faster memory state; compare lines of different boxes by it, report `value`")


def _finite(x):
    """NaN / inf (a stage event that was never recorded) -> None: the line must be strict JSON"""
    if isinstance(x, float):
        return x if math.isfinite(x) else None
    if isinstance(x, dict):
        return {k: _finite(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite(v) for v in x]
    return x


def frame_stats(torch, lib, vid, out_state, W, H, T, geom=None, P=0):
    """V, D, D_single, D_blend (and C, the cull pass's survivors) of the last foveated forward call (reads the workspaces)."""
    num_rendered, radii, img = out_state
    cands = 0
    if geom is not None:
        off = lib.fr_geometry_vis_count(vid, P, geom.data_ptr()) - geom.data_ptr()
        cands = int(geom[off:off + 4].view(torch.int32).item())
    rptr = lib.fr_image_ranges(vid, W, H, img.data_ptr())
    off = rptr - img.data_ptr()
    ranges = img[off:off + 8 * T].view(torch.int32).view(T, 2)
    lens = (ranges[:, 1] - ranges[:, 0]).to(torch.int64)
    lptr = lib.fr_image_tile_levels(W, H, img.data_ptr())
    off = lptr - img.data_ptr()
    lv = img[off:off + 20 * T].view(torch.float32).view(5, T)
    blend = lv[4] != 0
    d_blend = int(lens[blend].sum().item())
    d_all = int(lens.sum().item())
    return dict(V=int((radii > 0).sum().item()), C=cands, D=d_all, D_single=d_all - d_blend, D_blend=d_blend,
                max_list=int(lens.max().item()), blend_tiles=int(blend.sum().item()))


class load_profiles:
    """The committed rocprofv3 summaries (profiles/<tag>_pmc.json, <tag>_render_sq.json). PMC bytes are only quoted when
    the profile was made with this very library build (sha of libfovraster_hip.so recorded by tools/make_profiles.sh)."""
    STAGE_KERNELS = {"render": ["k_render_fov"], "project": ["k_project"], "bin": ["k_bin"],
                     "tile_sort": ["k_tile_msort", "k_tile_msort_direct", "k_split_long", "k_tile_msort_chunks"], "emit": ["k_emit"],
                     "tile_scan": ["k_tile_scan"], "tile_levels": ["k_tile_levels"]}

    def __init__(self):
        self.pmc = self.sq = None
        self.sha = lib_sha16()
        for name in ("pmc", "render_sq"):
            try:
                with open(os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_{name}.json")) as f:
                    setattr(self, "pmc" if name == "pmc" else "sq", json.load(f))
            except Exception:
                pass

    def traffic(self, stage, packed):
        if self.pmc is None:
            return None, "no profiles/%s_pmc.json" % PROFILE_TAG
        src = f"profiles/{PROFILE_TAG}_pmc.json (lib {self.pmc.get('lib_sha16')}, {self.pmc.get('layout')})"
        if self.pmc.get("lib_sha16") != self.sha:
            return None, src + f": made with another build of the library (this one is {self.sha}) -- not quoted"
        k = self.pmc["kernels"]
        f_fetch = self.pmc.get("calibration", {}).get("fetch_factor", 2.0)
        f_write = self.pmc.get("calibration", {}).get("write_factor", 1.0)
        try:
            names = [n + "_packed" if (packed and n + "_packed" in k) else n for n in self.STAGE_KERNELS[stage]]
            b = sum(f_fetch * k[n]["FETCH_SIZE_KiB_per_frame"] + f_write * k[n]["WRITE_SIZE_KiB_per_frame"] for n in names if n in k) * 1024
        except Exception:
            return None, src + ": kernel missing"
        return int(b), src + f"; bytes = {f_fetch} x FETCH_SIZE + {f_write} x WRITE_SIZE (factors calibrated on the k_pack_* launches of the same pass)"

    def train_source(self):
        if self.pmc is None or "train" not in self.pmc:
            return "no training-step PMC pass in profiles/%s_pmc.json" % PROFILE_TAG
        if self.pmc.get("lib_sha16") != self.sha:
            return f"profiles/{PROFILE_TAG}_pmc.json was made with another build ({self.pmc.get('lib_sha16')}, this one is {self.sha}) -- not quoted"
        return f"profiles/{PROFILE_TAG}_pmc.json [train] (lib {self.sha})"

    def traffic_train(self, kernel):
        """PMC bytes per launch of a kernel of the training step (the profile's `train` section: kernel names with their template
        arguments), or None."""
        if self.pmc is None or "train" not in self.pmc or self.pmc.get("lib_sha16") != self.sha:
            return None, None
        f_fetch = self.pmc.get("calibration", {}).get("fetch_factor", 2.0)
        f_write = self.pmc.get("calibration", {}).get("write_factor", 1.0)
        tot = 0.0
        for name, e in self.pmc["train"].items():
            base = name.split("<")[0]
            if base == kernel or (kernel.endswith("*") and base.startswith(kernel[:-1])):
                tot += (f_fetch * e.get("FETCH_SIZE_KiB_per_frame", 0) + f_write * e.get("WRITE_SIZE_KiB_per_frame", 0)) * 1024
        return (int(tot) if tot else None), None

    def blend_sq(self):
        if self.sq is None:
            return {}
        e = next((v for n, v in self.sq.get("kernels", {}).items() if n.startswith("k_render_fov")), None)
        if e is None:
            return {}
        out = {"sq_source": f"profiles/{PROFILE_TAG}_render_sq.json (lib {self.sq.get('lib_sha16')}" + ("" if self.sq.get("lib_sha16") == self.sha else f"; this build is {self.sha}") + ")"}
        for k in ("valu_frac", "occupancy_waves_per_simd", "wait_inst_frac", "wait_any_frac", "valu_insts", "lds_wait_frac"):
            if k in e:
                out[k] = e[k]
        return out


def extras(args, torch, np, syn, dev, cam, pc, cloud, bg, frame, render_plain, H, W):
    import torch.nn.functional as F
    from fov3dgs_amd.loss_utils import l1_ssim_loss
    extra = {}
    # --- the reference's FPS protocol (render_compose_gazes_fps.py:50-64): per gaze 10 warm-ups, then 5 renders with a
    # synchronize after each, events around the rasterizer call only; the reference's tensor interface
    starter, ender = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    per_gaze = {}
    with torch.no_grad():
        for packed, tag in ((None, "reference_protocol_fps"), ("auto", "reference_protocol_fps_packed")):
            fps_all = []
            for gaze in GAZES:
                for _ in range(10):
                    frame(gaze, packed, starter=starter, ender=ender)
                    torch.cuda.synchronize()
                ms = 0.0
                for _ in range(5):
                    frame(gaze, packed, starter=starter, ender=ender)
                    torch.cuda.synchronize()
                    ms += starter.elapsed_time(ender)
                fps_all.append(5 / (ms / 1000))
                if packed is None:
                    per_gaze[f"{gaze[0]:.2f},{gaze[1]:.2f}"] = round(fps_all[-1], 1)
            extra[tag] = round(float(np.mean(fps_all)), 2)
        extra["per_gaze_fps"] = per_gaze
        # --- moving gaze (Lissajous path, a new gaze every frame), wall clock
        for i in range(10):
            frame(syn.lissajous_gaze(i, 90), None)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(90):
            frame(syn.lissajous_gaze(10 + i, 90), None)
        torch.cuda.synchronize()
        extra["moving_gaze_fps"] = round(90 / (time.perf_counter() - t1), 2)
        # --- a workload of the SHAPE of the reference's one published number (fov3dgs/fps/ours-Q-9gazes/bicycle.txt:242: 702 FPS on
        # an unstated NVIDIA GPU): 1237 x 822, 1 161 358 Gaussians with the level counts of pnum/ours-Q/bicycle.txt:1-4 (the same
        # level fractions as the S-6M cloud), nine gazes, the reference's protocol. Synthetic cloud, other hardware: context only.
        from fov3dgs_amd.gaussian_renderer_fov import render as render_fov_small
        small_cpu = syn.scene_bicycle_scale(P=1_161_358, seed=1)
        fov_small = [t.to(dev) for t in syn.foveation_layers(small_cpu, seed=2)]
        small = small_cpu.to(dev)

        class _Frozen:
            pass
        ps = _Frozen()
        ps.get_xyz, ps.get_scaling = small.get_xyz.detach(), small.get_scaling.detach().contiguous()
        ps.get_rotation, ps.get_opacity = small.get_rotation.detach().contiguous(), small.get_opacity.detach().contiguous()
        ps.get_rest_features, ps.active_sh_degree = small.get_rest_features.detach().contiguous(), small.active_sh_degree
        cam_s = syn.camera_ring(0, 8, 1237, 822).to(dev)
        fps_all = []
        for gaze in GAZES:
            kw = dict(alpha=0.05, gazeArray=gaze, blending=True, highest_levels=fov_small[0], shs_dcs=fov_small[1], opacities=fov_small[2])
            for _ in range(10):
                render_fov_small(cam_s, ps, bg, starter=starter, ender=ender, **kw)
                torch.cuda.synchronize()
            ms = 0.0
            for _ in range(5):
                render_fov_small(cam_s, ps, bg, starter=starter, ender=ender, **kw)
                torch.cuda.synchronize()
                ms += starter.elapsed_time(ender)
            fps_all.append(5 / (ms / 1000))
        extra["reference_shaped_fps"] = round(float(np.mean(fps_all)), 1)
        extra["reference_shaped_note"] = ("1237x822, 1 161 358 Gaussians (level fractions of pnum/ours-Q/bicycle.txt), nine gazes, 10 warm-ups + 5 renders "
                                          "each, events around the rasterizer: the shape of the reference's published 702 FPS (fps/ours-Q-9gazes/"
                                          "bicycle.txt:242, unstated NVIDIA GPU, the real bicycle model) -- synthetic cloud, other hardware: context only")
        del small, small_cpu, fov_small, ps
        # --- non-foveated forward (config 2)
        for _ in range(3):
            render_plain(cam, pc, Pipe(), bg, cuda_type="pcheck_obb")
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(20):
            render_plain(cam, pc, Pipe(), bg, cuda_type="pcheck_obb")
        torch.cuda.synchronize()
        extra["nonfov_forward_fps"] = round(20 / (time.perf_counter() - t1), 2)
    # --- training step (config 4): eff_finetune.py's render -> 0.8 L1 + 0.2 (1 - SSIM) -> backward, no optimizer;
    # forward / loss / backward timed apart with events, median of 50
    tr = cloud.requires_grad_(True)
    target = torch.rand(3, H, W, device=dev)
    extra.update(training_extras(torch, np, syn, render_plain, cam, bg, tr, target, H, W))
    # --- roofline of the training step's kernels (BASELINE metric "fwd+bwd ms/iter; HBM GB/s"): a short instrumented loop of the
    # raw-parameter step -- HIP events around the forward stages and around the backward pass's kernels, recorded by the library on
    # the streams the kernels run on; algorithmic bytes: training_bytes(); PMC bytes: the committed profile of the training step
    from fov3dgs_amd.profiling import BackwardTimer, StageTimer
    from fov3dgs_amd import _native as _nat
    tr.fuse_activations, tr.row_sparse_grads = True, False
    n_inst = 12
    from fov3dgs_amd.profiling import NativeCallTimer
    ft, bt, nt = StageTimer(n_inst), BackwardTimer(n_inst), NativeCallTimer()
    with ft, bt, nt:
        for it in range(n_inst):
            for p in tr.parameters():
                p.grad = None
            o = render_plain(cam, tr, Pipe(), bg, cuda_type="pcheck_obb_sum")
            l1_ssim_loss(o["render"], target, 0.2).backward()
    torch.cuda.synchronize()
    fwd_ms = {k: float(np.median([r_[k] for r_ in ft.stage_ms()[2:]])) for k in _nat.STAGES}
    bwd_rows = bt.stage_ms()[2:]
    bwd_ms = {k: float(np.median([r_[k] for r_ in bwd_rows])) for k in ("render_bwd", "preprocess_bwd", "fill_zero")}
    # the gradient tensors' zero fill: by default inside fr_backward, on the helper stream beside k_render_bwd (events 3 / 4 of the call);
    # with rasterizer.PREZERO_GRADIENTS at the end of the forward call on a side stream (its own duration from events on that stream)
    fills = nt.ms()["fill"]
    if len(fills) > 2:
        bwd_ms["fill_zero"] = float(np.median(fills[2:]))
    ft.close(); bt.close()
    with torch.no_grad():
        vmat = cam.world_view_transform
        v_in = int(((tr.get_xyz @ vmat[:3, 2] + vmat[3, 2]) > 0.2).sum().item())
        from fov3dgs_amd import rasterizer as rz_
        rs_ = rz_.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg, 1.0, cam.world_view_transform,
                                                cam.full_proj_transform, 3, cam.camera_center, False, False)
        T_ = ((W + 15) // 16) * ((H + 15) // 16)
        pf, pb = torch.zeros(T_, dtype=torch.int32, device=dev), torch.zeros(T_, dtype=torch.int32, device=dev)
        r_tr = rz_._forward_native(_nat.VARIANT_PCHECK_OBB_SUM, rs_, pc.get_xyz, pc.get_features, torch.Tensor([]), pc.get_opacity,
                                   pc.get_scaling, pc.get_rotation, torch.Tensor([]), blend_pairs=pf)
        d_tr = r_tr[0]  # the frame's instances
        E_ = torch.Tensor([])
        rz_._backward_native(_nat.VARIANT_PCHECK_OBB_SUM, rs_, pc.get_xyz, r_tr[2], E_, pc.get_opacity, pc.get_scaling, pc.get_rotation, E_,
                             torch.rand(3, H, W, device=dev), pc.get_features, r_tr[3], d_tr, r_tr[4], r_tr[5], blend_pairs=pb)
        torch.cuda.synchronize()
    n_tr = dict(P=int(tr.get_xyz.shape[0]), Px=H * W, T=((W + 15) // 16) * ((H + 15) // 16), V=int(o["visibility_filter"].sum().item()),
                V_in=v_in, D=int(d_tr), blend_pairs_fwd=int(pf.long().sum().item()), blend_pairs_bwd=int(pb.long().sum().item()))
    extra["_train_counts"], extra["_train_fwd_ms"], extra["_train_bwd_ms"] = n_tr, fwd_ms, bwd_ms
    tr.row_sparse_grads = False
    return extra


def reference_loss_fn(torch, dev):
    """The reference's formulation of the training loss (fov3dgs/utils/loss_utils.py:17-76 as eff_finetune.py:124-125 combines it:
    l1_loss + ssim = five grouped conv2d's with an 11 x 11 Gaussian window + elementwise ops, differentiated by autograd), in torch."""
    import torch.nn.functional as F
    g1 = torch.tensor([math.exp(-(i - 5) ** 2 / 4.5) for i in range(11)], device=dev)
    g1 = g1 / g1.sum()
    win = (g1[:, None] @ g1[None, :])[None, None].expand(3, 1, 11, 11).contiguous()

    def torch_loss(img, gt, lambda_dssim=0.2):
        a, b = img[None], gt[None]
        mu1, mu2 = F.conv2d(a, win, padding=5, groups=3), F.conv2d(b, win, padding=5, groups=3)
        s1 = F.conv2d(a * a, win, padding=5, groups=3) - mu1 * mu1
        s2 = F.conv2d(b * b, win, padding=5, groups=3) - mu2 * mu2
        s12 = F.conv2d(a * b, win, padding=5, groups=3) - mu1 * mu2
        m = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))
        return (1.0 - lambda_dssim) * (img - gt).abs().mean() + lambda_dssim * (1.0 - m.mean())
    return torch_loss


def train_step_times(torch, np, render_plain, cam, bg, model, params, target, loss_fn, n=50, want_stats=True):
    """n + 5 steps of render(pcheck_obb_sum) -> loss -> backward (no optimizer), back to back as a training loop runs them (the
    forward call's wait for its instance count is the only synchronisation). -> dict: fwd / loss_fwd / bwd = medians of the
    events around the three phases on the stream; step = wall clock of the n steps / n; raster_fwd / raster_bwd = medians of the
    events around the native fr_forward / fr_backward calls alone (rasterizer-only)."""
    from fov3dgs_amd.profiling import NativeCallTimer
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(n + 5)]
    t1 = 0.0
    kw = {} if want_stats else {"want_stats": False}
    with NativeCallTimer() as nt:
        for it in range(n + 5):
            if it == 5:
                torch.cuda.synchronize()
                t1 = time.perf_counter()
            ev = evs[it]
            for p in params:
                p.grad = None
            ev[0].record()
            o = render_plain(cam, model, Pipe(), bg, cuda_type="pcheck_obb_sum", **kw)
            ev[1].record()
            loss = loss_fn(o["render"], target, 0.2)
            ev[2].record()
            loss.backward()
            ev[3].record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t1) / n * 1e3
        call = nt.ms()
    rows = np.array([(e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), e[2].elapsed_time(e[3])) for e in evs[5:]])
    med = np.median(rows, axis=0)
    return dict(fwd=round(float(med[0]), 3), loss_fwd=round(float(med[1]), 3), bwd=round(float(med[2]), 3), step=round(wall, 3),
                raster_fwd=round(float(np.median(call["fwd"][5:])), 3), raster_bwd=round(float(np.median(call["bwd"][5:])), 3)), o


def training_extras(torch, np, syn, render_plain, cam, bg, tr, target, H, W, short=False):
    """The training step of BASELINE config 4 in every form this package offers, on the model `tr` (requires_grad). Keys:
      train_ref_*        a model that exposes ONLY the reference's getters (scene/gaussian_model.py:200-240: exp / normalize / sigmoid as
                         torch expressions, get_features = torch.cat), dense gradients, the reference's torch L1 + SSIM
                         (utils/loss_utils.py): what a maintainer gets from switching the imports, nothing else
      train_ref_fused_*  the same model with this package's fused L1 + SSIM (one more import)
      train_refmodel_*   a model with the reference GaussianModel's ATTRIBUTES too (raw tensors + torch.exp / sigmoid / normalize as
                         scaling_ / opacity_ / rotation_activation): render() recognises it and skips the getters' temporaries
                         (gaussian_renderer.FAST_REFERENCE_MODEL); fused loss
      train_*            + the three activations as one fused pass each way, split SH storage (extension getters)
      train_raw_*        + the raw parameters handed to the rasterizer (activations inside its kernels)
      train_raw_nostats_* + render(want_stats=False): no gs_count / contribs (eff_finetune.py:107-108 drops them)
      train_sparse_*     raw parameters + row-sparse gradients (the reference's contract is dense)
    each: fwd / loss_fwd / bwd (events on the stream, median), step (wall clock of back-to-back steps), raster_fwd / raster_bwd
    (events around the native calls alone)."""
    from fov3dgs_amd.loss_utils import l1_ssim_loss
    dev = target.device
    torch_loss = reference_loss_fn(torch, dev)
    n = 20 if short else 50
    out = {}
    params = tr.parameters()
    ref = syn.ReferenceGetterModel(tr)

    def put(prefix, model, loss_fn, **kw):
        t, o = train_step_times(torch, np, render_plain, cam, bg, model, params, target, loss_fn, n=n, **kw)
        out[prefix + "fwd_ms"], out[prefix + "loss_fwd_ms"], out[prefix + "bwd_ms"], out[prefix + "step_ms"] = t["fwd"], t["loss_fwd"], t["bwd"], t["step"]
        out[prefix + "raster_fwd_ms"], out[prefix + "raster_bwd_ms"] = t["raster_fwd"], t["raster_bwd"]
        return o
    tr.fuse_activations, tr.row_sparse_grads = False, False
    put("train_ref_", ref, torch_loss)
    put("train_ref_fused_", ref, l1_ssim_loss)
    put("train_refmodel_", syn.ReferenceShapedModel(tr), l1_ssim_loss)
    if not short:
        put("train_", tr, l1_ssim_loss)
    tr.fuse_activations = True
    put("train_raw_", tr, l1_ssim_loss)
    put("train_raw_nostats_", tr, l1_ssim_loss, want_stats=False)
    if not short:
        tr.row_sparse_grads = True
        put("train_sparse_", tr, l1_ssim_loss)
        tr.row_sparse_grads = False
    out["train_note"] = training_extras.__doc__.split("Keys:")[1].strip()
    return out


def multi_gpu_extras(args, rank, world, dev, cloud, cam, bg, multiview, render_plain, barrier_sync, frame, torch, np):
    """N > 1, render mode: the views are independent, so the headline has no collective in it. What north_star's exchange
    steps cost is measured here, untimed for the headline: (a) the same frames with every rank's image gathered on rank 0
    (asynchronous RCCL gather overlapped with the next frame), (b) a few multi-view training steps (forward + fused loss +
    backward per rank, then the gradient sum over ranks). -> dict for `extra` (rank 0 holds the maxima over ranks)."""
    from fov3dgs_amd.loss_utils import l1_ssim_loss
    out = {}
    K = min(args.steps, 27)
    with torch.no_grad():
        pending = None
        for i in range(3):
            multiview.gather_images(frame(GAZES[i % 9], None)["render"], dst=0)
        barrier_sync()
        t0 = time.perf_counter()
        for i in range(K):
            o = frame(GAZES[i % 9], None)
            if pending is not None:
                pending[0].wait()
            pending = multiview.gather_images(o["render"], dst=0, async_op=True) + (o["render"],)
        pending[0].wait()
        barrier_sync()
        el = time.perf_counter() - t0
    tr = cloud.requires_grad_(True)
    tr.fuse_activations = True
    H, W = args.height, args.width
    target = torch.rand(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(100 + rank))
    params = list(tr.parameters())
    n_train = 6
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n_train)]
    info = None
    for it in range(2 + n_train):
        for p_ in params:
            p_.grad = None
        ev = evs[it - 2] if it >= 2 else None
        if ev:
            ev[0].record()
        o = render_plain(cam, tr, Pipe(), bg, cuda_type="pcheck_obb_sum")
        l1_ssim_loss(o["render"], target, 0.2).backward()
        if ev:
            ev[1].record()
        info = multiview.allreduce_gradients(params, visible=o["visibility_filter"])
        if ev:
            ev[2].record()
    barrier_sync()
    fb = float(np.median([e[0].elapsed_time(e[1]) for e in evs]))
    co = float(np.median([e[1].elapsed_time(e[2]) for e in evs]))
    t = torch.tensor([el, fb, co], device=dev, dtype=torch.float64)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    el, fb, co = [float(x) for x in t.tolist()]
    # the same steps with the dense exchange started INSIDE the backward pass (multiview.OverlappedGradientExchange): the step's
    # forward-to-summed-gradients time and what of the exchange stayed exposed behind the backward pass's own kernels
    overlapped = None
    try:
        named = {"means3D": tr._xyz, "opacities": tr._opacity, "scales": tr._scaling, "rotations": tr._rotation, "sh": tr._features_dc, "sh_rest": tr._features_rest}
        oev, exps = [], []
        for it in range(2 + n_train):
            for p_ in params:
                p_.grad = None
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            o = render_plain(cam, tr, Pipe(), bg, cuda_type="pcheck_obb_sum")
            loss = l1_ssim_loss(o["render"], target, 0.2)
            ex = multiview.OverlappedGradientExchange(named, ranges=4)
            with ex:
                loss.backward()
            e1.record()
            if it >= 2:
                oev.append((e0, e1))
                exps.append(ex)
        barrier_sync()
        t2 = torch.tensor([float(np.median([a_.elapsed_time(b_) for a_, b_ in oev])), float(np.median([x_.exposed_ms() or 0.0 for x_ in exps]))],
                          device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t2, op=torch.distributed.ReduceOp.MAX)
        overlapped = dict(step_ms=round(float(t2[0]), 4), collective_exposed_ms=round(float(t2[1]), 4), ranges=4)
    except Exception as e:  # (a diagnosis must not take the line down)
        overlapped = f"failed: {e}"
    for p_ in params:
        p_.grad = None
    tr.requires_grad_(False)
    nbytes = (info or {}).get("bytes", 0)
    algbw = nbytes / (co * 1e-3) / 1e9 if co > 0 else None
    out["gather_fps"] = round(world * K / el, 2)
    out["gather_note"] = f"{K} frames per rank, every frame's image gathered on rank 0 (asynchronous gather of the previous frame under the next one)"
    # diagnosis for the first real multi-GPU run: what RCCL looks like from rank 0, the dense exchange tensor by tensor (the
    # 1.15 GB dL_dsh all-reduce dominates), and the same sum as one flat reduce-scatter + all-gather
    out["rccl"] = multiview.comm_info()
    try:
        for p_ in params:
            p_.grad = torch.zeros_like(p_)
        per_t = []
        multiview.allreduce_gradients(params, per_tensor_ms=per_t)
        tm = {}
        multiview.allreduce_gradients_flat(params, timings=tm)
        out["collective_breakdown"] = dict(per_tensor=[dict(MB=round(n_ * 4 / 1e6, 1), ms=round(ms_, 3)) for n_, ms_ in per_t],
                                           flat_reduce_scatter_all_gather_ms={k_: round(v_, 3) for k_, v_ in tm.items()})
        for p_ in params:
            p_.grad = None
    except Exception as e:  # (gloo on a shared GPU has no reduce_scatter_tensor: the headline must not die of a diagnosis)
        out["collective_breakdown"] = f"failed: {e}"
    out["multiview_train"] = dict(fwd_bwd_ms=round(fb, 4), collective_ms=round(co, 4), backend=torch.distributed.get_backend(),
                                  collective=dict(info or {}, algbw_GBs=None if algbw is None else round(algbw, 2),
                                                  busbw_GBs=None if algbw is None else round(algbw * 2 * (world - 1) / world, 2)),
                                  overlapped_exchange=overlapped,
                                  note=f"median of {n_train} steps: pcheck_obb_sum forward + fused L1+SSIM + backward of this rank's camera, then "
                                       "multiview.allreduce_gradients over the ranks (max over ranks); overlapped_exchange: the dense sum started inside the "
                                       "backward pass (step_ms = forward to summed gradients; fwd_bwd_ms + collective_ms is the same span without it)")
    return out


def train_mode(args, rank, world, dev, cloud, cam, bg, multiview, render_plain, barrier_sync):
    """Config 5: one camera per rank, forward + loss + backward, then the gradient sum over ranks."""
    import numpy as np
    import torch
    from fov3dgs_amd.loss_utils import l1_ssim_loss
    K, Wm = args.steps, args.warmup
    H, W = args.height, args.width
    tr = cloud.requires_grad_(True)
    tr.fuse_activations = not args.activation_pass
    tr.row_sparse_grads = bool(args.row_sparse)
    target = torch.rand(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(100 + rank))
    params = list(tr.parameters())
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(K)]
    info = None

    def step(ev):
        nonlocal info
        for p in params:
            p.grad = None
        if ev:
            ev[0].record()
        o = render_plain(cam, tr, Pipe(), bg, cuda_type="pcheck_obb_sum")
        l1_ssim_loss(o["render"], target, 0.2).backward()
        if ev:
            ev[1].record()
        info = multiview.allreduce_gradients(params, visible=o["visibility_filter"])
        if ev:
            ev[2].record()
    for _ in range(Wm):
        step(None)
    barrier_sync()
    t0 = time.perf_counter()
    for i in range(K):  # back to back, as a training loop runs them: no synchronisation between steps
        step(evs[i])
    barrier_sync()
    elapsed = time.perf_counter() - t0
    fb = [e[0].elapsed_time(e[1]) for e in evs]
    co = [e[1].elapsed_time(e[2]) for e in evs]
    # the same steps with the dense exchange STARTED INSIDE the backward pass (multiview.OverlappedGradientExchange: the per-Gaussian
    # pass in four ranges of rows, each range's all-reduce on a communication stream as soon as it is complete): the step's wall
    # clock and what of the exchange stayed exposed behind the backward pass's own kernels
    overlapped = None
    if world > 1 and tr.fuse_activations and not args.row_sparse:
        named = {"means3D": tr._xyz, "opacities": tr._opacity, "scales": tr._scaling, "rotations": tr._rotation, "sh": tr._features_dc, "sh_rest": tr._features_rest}
        exposed = []

        def ostep(timed):
            for p in params:
                p.grad = None
            o = render_plain(cam, tr, Pipe(), bg, cuda_type="pcheck_obb_sum")
            loss = l1_ssim_loss(o["render"], target, 0.2)
            ex = multiview.OverlappedGradientExchange(named, ranges=4)
            with ex:
                loss.backward()
            if timed:
                exposed.append(ex)
        for _ in range(max(Wm, 1)):
            ostep(False)
        barrier_sync()
        t1 = time.perf_counter()
        for i in range(K):
            ostep(True)
        barrier_sync()
        el_o = time.perf_counter() - t1
        exp_ms = float(np.median([e.exposed_ms() or 0.0 for e in exposed]))
        t = torch.tensor([el_o, exp_ms], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        overlapped = dict(ms_per_step=round(float(t[0]) / K * 1e3, 4), collective_exposed_ms=round(float(t[1]), 4), ranges=4,
                          note="dense all-reduce of every range of rows started behind the k_preprocess_bwd launch that completes it; "
                               "collective_exposed_ms = from the end of the backward pass's kernels to the end of the exchange, on the compute stream")
    if world > 1:
        t = torch.tensor([elapsed, float(np.median(fb)), float(np.median(co))], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed, fbm, com = [float(x) for x in t.tolist()]
    else:
        fbm, com = float(np.median(fb)), float(np.median(co))
    if rank != 0:
        return
    nbytes = (info or {}).get("bytes", 0)
    algbw = nbytes / (com * 1e-3) / 1e9 if com > 0 and world > 1 else None
    line = {"metric": "fwd+bwd ms/iter (multi-view training step, one camera per GPU)", "value": round(elapsed / K * 1e3, 4), "unit": "ms/iter",
            "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": False,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "S-6M cloud, pcheck_obb_sum forward + fused 0.8 L1 + 0.2 (1 - SSIM) + backward per rank (model "
                                   "activations " + ("inside the rasterizer's kernels" if tr.fuse_activations else "as one fused pass each way")
                                   + (", row-sparse gradients" if args.row_sparse else "") + "), gradient sum over ranks (multiview.allreduce_gradients)", "gaussians": args.points, "width": W, "height": H,
                       "parallelism": f"views{world}"},
            "fwd_bwd_ms": round(fbm, 4), "collective_ms": round(com, 4), "views_per_s": round(world * K / elapsed, 3),
            "overlapped_exchange": overlapped, "collective_exposed_ms": None if overlapped is None else overlapped["collective_exposed_ms"],
            "collective": dict(info or {}, algbw_GBs=None if algbw is None else round(algbw, 2),
                               busbw_GBs=None if algbw is None else round(algbw * 2 * (world - 1) / world, 2))}
    print(json.dumps(_finite(line)), flush=True)


def cpu_baseline(cloud_cpu, fov_cpu, cam, gaze, T_tiles, gx, gy):
    """The CPU oracle (C port of the reference algorithm, OpenMP over Gaussians and tiles) on all host cores. Every
    Gaussian is preprocessed and culled; binning + sort + blending run over the whole frame when a 32x32-tile probe says
    that fits the time budget, else over the probe window and extrapolated (said so in `sample`)."""
    from oracle import oracle as orc
    from tests.helpers import cam_dict, scene_dict
    cores = os.cpu_count() or 1
    orc.set_threads(cores)
    scene = scene_dict(cloud_cpu, "fov_pcheck_obb", fov_cpu)
    wx, wy = min(32, gx), min(32, gy)
    x0, y0 = (gx - wx) // 2, (gy - wy) // 2
    cd = cam_dict(cam, gaze=gaze, alpha=0.05)
    cd["capacity_hint"] = 16_000_000

    def timed(win, reps=2):
        cd["tile_window"] = win
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            orc.forward("fov_pcheck_obb", scene, cd)
            best = min(best, time.perf_counter() - t0)
        return best
    t_total0 = time.perf_counter()
    t_win = timed((x0, y0, x0 + wx, y0 + wy))
    t_one = timed((x0 + wx // 2, y0 + wy // 2, x0 + wx // 2 + 1, y0 + wy // 2 + 1))
    per_tile = max(t_win - t_one, 0.0) / (wx * wy - 1)
    est = t_one + per_tile * (T_tiles - 1)
    if est < 12.0:
        t_frame = timed(None, reps=2)
        how = f"whole frame ({T_tiles} tiles) binned / sorted / blended: {t_frame:.2f} s"
    else:
        t_frame = est
        how = (f"{wx}x{wy}-tile centre window binned / sorted / blended in {t_win:.2f} s, 1x1 window {t_one:.2f} s; frame time "
               f"EXTRAPOLATED to {T_tiles} tiles = {t_frame:.2f} s")
    return dict(value=round(1.0 / t_frame, 5), unit="frames/s", cores=cores, kind="port",
                sample=f"oracle/fovraster_oracle.c (C port, OpenMP, {cores} threads), centre gaze: all {len(cloud_cpu)} Gaussians preprocessed + culled; " + how,
                seconds_measured=round(time.perf_counter() - t_total0, 2))


if __name__ == "__main__":
    main()
    try:
        import torch
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
    except Exception:
        pass
