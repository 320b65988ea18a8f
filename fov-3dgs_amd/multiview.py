"""Multi-view data parallelism: one process per GPU, one camera per rank.

The reference has no distributed code; a multi-view batch shards naturally across views because
every camera's render (and backward) is independent given the same, replicated Gaussian cloud.
The only exchanges are an image gather after the forward pass and a gradient sum after the
backward pass -- RCCL (`backend="nccl"` on ROCm) over xGMI on GPUs, gloo in the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def init_distributed():
    """-> (rank, world_size, local_rank). Reads the torchrun environment; no-op for a single process.

    The backend must come out the same on every rank, so it is decided from facts all ranks of a node share: RCCL
    ("nccl") when every local rank has a GPU of its own (ranks per node <= devices; the ranks per node default to
    WORLD_SIZE when the launcher does not say), gloo when ranks share devices (a test box: RCCL refuses two ranks on one
    GPU) or there is no GPU at all."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    have_gpu = torch.cuda.is_available()
    ndev = torch.cuda.device_count() if have_gpu else 0
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    shared = have_gpu and local_world > ndev
    if have_gpu:
        local_rank %= ndev
    if world > 1 and not dist.is_initialized():
        if have_gpu:
            torch.cuda.set_device(local_rank)
        if have_gpu and not shared:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    return rank, world, local_rank


def views_for_rank(rank, world, n_views):
    """Indices of the cameras rank `rank` renders when `n_views` cameras are split over `world` ranks
    (contiguous blocks; with n_views == world this is one camera per GPU)."""
    per = (n_views + world - 1) // world
    return list(range(rank * per, min(n_views, (rank + 1) * per)))


def gather_images(image, dst=0, async_op=False):
    """Collect every rank's [3,H,W] image on `dst`. Returns (work_or_None, list_or_None)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return None, [image]
    world = dist.get_world_size()
    out = [torch.empty_like(image) for _ in range(world)] if dist.get_rank() == dst else None
    work = dist.gather(image, out, dst=dst, async_op=async_op)
    return (work if async_op else None), out


def all_gather_images(image, async_op=False):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return None, image.unsqueeze(0)
    world = dist.get_world_size()
    # concatenated layout [world*C, H, W] (accepted by both RCCL and gloo), viewed as [world, C, H, W]
    flat = torch.empty((world * image.shape[0],) + tuple(image.shape[1:]), dtype=image.dtype, device=image.device)
    work = dist.all_gather_into_tensor(flat, image.contiguous(), async_op=async_op)
    return (work if async_op else None), flat.view((world,) + tuple(image.shape))


def comm_info():
    """What the process group looks like from this rank (for the bench line: was RCCL really running with N ranks?)."""
    if not dist.is_initialized():
        return dict(backend=None, world=1)
    info = dict(backend=dist.get_backend(), world=dist.get_world_size(), rank=dist.get_rank(),
                local_world=int(os.environ.get("LOCAL_WORLD_SIZE", str(dist.get_world_size()))),
                device_count=torch.cuda.device_count() if torch.cuda.is_available() else 0)
    try:
        v = torch.cuda.nccl.version()  # RCCL's version on ROCm
        info["nccl_version"] = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:
        info["nccl_version"] = None
    return info


def allreduce_gradients_flat(params, timings=None):
    """The dense gradient sum as ONE flat reduce-scatter + all-gather over a single buffer (the two halves of a ring all-reduce,
    exposed: with N ranks every rank sums 1/N of the 1.42 GB and the halves run over all xGMI links at once) instead of one
    all-reduce per tensor. Costs a pack and an unpack pass over the gradients (2 x 1.42 GB of HBM traffic at 6 M Gaussians,
    ~0.5 ms); opt-in, for comparison with the per-tensor exchange on the 8-GPU node (single ring ~16 ms vs all
    links ~3 ms). timings: optional dict that receives per-phase CUDA-event milliseconds (pack, reduce_scatter, all_gather, unpack)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return dict(mode="none", bytes=0)
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return dict(mode="none", bytes=0)
    world = dist.get_world_size()
    n = sum(g.numel() for g in grads)
    per = (n + world - 1) // world
    flat = torch.zeros(per * world, dtype=grads[0].dtype, device=grads[0].device)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if (timings is not None and flat.is_cuda) else None
    if ev:
        ev[0].record()
    off = 0
    for g in grads:
        flat[off:off + g.numel()].copy_(g.reshape(-1))
        off += g.numel()
    if ev:
        ev[1].record()
    shard = torch.empty(per, dtype=flat.dtype, device=flat.device)
    dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM)
    if ev:
        ev[2].record()
    dist.all_gather_into_tensor(flat, shard)
    if ev:
        ev[3].record()
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()
    if ev:
        ev[4].record()
        torch.cuda.synchronize()
        for name, i in (("pack", 0), ("reduce_scatter", 1), ("all_gather", 2), ("unpack", 3)):
            timings[name] = ev[i].elapsed_time(ev[i + 1])
    return dict(mode="flat", bytes=n * flat.element_size())


def _allreduce_sparse_rows(params):
    """Gradients that already ARE row-sparse (the rasterizer's row_sparse extension: torch.sparse_coo with one sparse dimension,
    the Gaussians this rank's view touched): exchanged as the rows of the union of the ranks' row sets -- no dense [P, ...]
    tensor is ever made. p.grad becomes the summed sparse tensor over the union's rows (the same rows on every rank)."""
    grads = [p.grad.coalesce() for p in params]
    P, dev = grads[0].shape[0], grads[0].device
    union = torch.zeros(P, dtype=torch.uint8, device=dev)
    for g in grads:
        union[g.indices()[0]] = 1
    dist.all_reduce(union, op=dist.ReduceOp.MAX)
    idx = torch.nonzero(union, as_tuple=False).squeeze(1)
    U = int(idx.numel())
    widths = [g.values()[0].numel() if g.values().shape[0] else int(torch.tensor(g.shape[1:]).prod().item()) for g in grads]
    rows = torch.zeros((U, sum(widths)), dtype=grads[0].dtype, device=dev)
    off = 0
    for g, w in zip(grads, widths):
        pos = torch.searchsorted(idx, g.indices()[0])
        rows[pos, off:off + w] = g.values().reshape(-1, w)
        off += w
    dist.all_reduce(rows, op=dist.ReduceOp.SUM)
    off = 0
    for p_, g, w in zip(params, grads, widths):
        p_.grad = torch.sparse_coo_tensor(idx.unsqueeze(0), rows[:, off:off + w].reshape((U,) + tuple(g.shape[1:])), g.shape, is_coalesced=True)
        off += w
    return dict(mode="sparse_rows", rows=U, of=P, bytes=rows.numel() * rows.element_size() + P)


def allreduce_gradients(params, visible=None, sparse_below=0.4, check_rows=False, per_tensor_ms=None):
    """Sum the per-view gradients of the replicated parameters over all ranks. -> dict with what was exchanged.

    dense (default): every gradient tensor is all-reduced IN PLACE, all collectives in flight at once (no flat copy: the
    previous torch.cat + copy-back cost two extra passes over 1.42 GB at 6 M Gaussians; RCCL chunks a 1.15 GB tensor
    by itself, and xGMI rings are per-link bound, so few large collectives are the right shape).
    row-sparse (`visible` = this rank's bool [P] mask of the Gaussians its view touched, i.e. radii > 0): a Gaussian no
    view sees has an all-zero gradient row on every rank, so only the rows of the UNION of the masks are exchanged --
    one 1-byte-per-Gaussian MAX all-reduce for the union, a gather of those rows from every [P, ...] gradient into one
    [U, row] buffer, one all-reduce, a scatter back. Used when the union covers less than `sparse_below` of the
    Gaussians (two or three views of a large scene; eight views on a ring see most of it, and dense wins).
    PRECONDITION of the row exchange (passing `visible` opts in): rows outside a rank's mask are ZERO on that rank, i.e.
    p.grad was cleared before the step and holds nothing but this view's rasterizer gradients. With gradient
    accumulation or a dense loss term (opacity / scale regularisers, a mask loss) rows outside the union would be left
    un-summed -- use the dense exchange (visible=None) there. check_rows=True verifies the precondition (one reduction
    per tensor and a host sync: for tests and debugging). per_tensor_ms: optional list that receives (numel, milliseconds) of
    every tensor's all-reduce, issued one after the other and timed with CUDA events (diagnosis: which tensor dominates)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return dict(mode="none", bytes=0)
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return dict(mode="none", bytes=0)
    P = grads[0].shape[0]
    if all(g.is_sparse for g in grads):
        return _allreduce_sparse_rows([p for p in params if p.grad is not None])
    if visible is not None and all(g.shape[0] == P for g in grads):
        union = visible.to(torch.uint8)
        dist.all_reduce(union, op=dist.ReduceOp.MAX)
        idx = torch.nonzero(union, as_tuple=False).squeeze(1)
        U = int(idx.numel())
        if U < sparse_below * P:
            if check_rows:
                mine = visible.reshape(P).to(torch.bool)
                for g in grads:
                    if bool((g.reshape(P, -1)[~mine] != 0).any()):
                        raise RuntimeError("allreduce_gradients(visible=...): a gradient row outside this rank's visibility mask is not "
                                           "zero (gradient accumulation or a dense loss term?) -- use the dense exchange")
            widths = [g[0].numel() for g in grads]
            rows = torch.empty((U, sum(widths)), dtype=grads[0].dtype, device=grads[0].device)
            off = 0
            for g, w in zip(grads, widths):
                rows[:, off:off + w] = g.reshape(P, w).index_select(0, idx)
                off += w
            dist.all_reduce(rows, op=dist.ReduceOp.SUM)
            off = 0
            for g, w in zip(grads, widths):
                g.reshape(P, w).index_copy_(0, idx, rows[:, off:off + w])
                off += w
            return dict(mode="rows", rows=U, of=P, bytes=rows.numel() * rows.element_size() + P)
    if per_tensor_ms is not None and grads[0].is_cuda:
        evs = []
        for g in grads:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
            e1.record()
            evs.append((g.numel(), e0, e1))
        torch.cuda.synchronize()
        per_tensor_ms.extend((n_, e0.elapsed_time(e1)) for n_, e0, e1 in evs)
        return dict(mode="dense", bytes=sum(g.numel() * g.element_size() for g in grads))
    works = [dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True) for g in grads]
    for w in works:
        w.wait()
    return dict(mode="dense", bytes=sum(g.numel() * g.element_size() for g in grads))


class OverlappedGradientExchange:
    """The dense gradient sum of a multi-view training step, STARTED INSIDE the backward pass (round 6; SURVEY.md 8e: the exchange moves
    1.42 GB per rank at 6 M Gaussians, 2.5-3 ms over xGMI against a 2.3 ms step, and `allreduce_gradients` only starts it when
    backward() has returned).

        ex = OverlappedGradientExchange({"means3D": pc._xyz, "opacities": pc._opacity, "scales": pc._scaling, "rotations": pc._rotation,
                                         "sh": pc._features_dc, "sh_rest": pc._features_rest})
        with ex:
            loss.backward()
        # here every p.grad holds the sum over the ranks; ex.exposed_ms() = what of the exchange was NOT hidden behind the backward pass

    How: the rasterizer runs the per-Gaussian half of its backward call (k_preprocess_bwd: 40 % of the call) in `ranges` pieces over
    increasing ranges of rows and tells this object behind each piece (rasterizer.GRADIENT_RANGE_HOOK; C ABI:
    fr_backward_args.num_ranges / range_done). Every gradient tensor's rows of that range are then final on the compute stream, zeros
    included, and their all-reduce goes out on a communication stream that waits for exactly that point: all but the last range's
    share of the exchange runs beside the rest of the backward pass (RCCL's own kernels take a few CUs; the per-Gaussian pass is bound
    by HBM's random-row rate, not by CUs).
    For models whose rasterizer inputs ARE the leaf parameters (raw activations + split SH storage: the reference-shaped model that
    render() recognises, GaussianCloud with fuse_activations): the gradient tensors of the call then become p.grad as they are. The
    keys name the rasterizer's inputs; a parameter may be missing (None). PRECONDITION: p.grad is None on entry (no accumulation: the
    sum is formed in the tensors the backward call allocates). If autograd copied a tensor instead of adopting it, p.grad is set
    to the summed tensor on exit. One backward call of one rasterizer per `with` block. Works on CPU tensors / gloo as well
    (synchronous collectives: the tests)."""

    def __init__(self, params, ranges=4, group=None):
        self.params = {k: v for k, v in params.items() if v is not None}
        self.ranges, self.group = int(ranges), group
        self.comm = None
        self._tensors, self._works, self._events = None, [], None
        self.calls = []

    def _active(self):
        return dist.is_initialized() and dist.get_world_size(self.group) > 1

    def __enter__(self):
        from . import rasterizer as rz
        for k, p_ in self.params.items():
            if p_.grad is not None:
                raise RuntimeError(f"OverlappedGradientExchange: {k}.grad must be None on entry (the sum is formed in the backward call's own tensors)")
        self._tensors, self._works, self.calls = None, [], []
        if self._active():
            self._saved = (rz.GRADIENT_RANGE_HOOK, rz.GRADIENT_RANGES)
            rz.GRADIENT_RANGE_HOOK, rz.GRADIENT_RANGES = self._on_range, self.ranges
        return self

    def _on_range(self, k, lo, hi, grads):
        """rows [lo, hi) of every tensor of `grads` are complete on the current stream once it gets here: sum them over the ranks"""
        if self._tensors is None:
            self._tensors = grads
        elif self._tensors is not grads and k == 0:
            raise RuntimeError("OverlappedGradientExchange: a second backward call inside one `with` block")
        self.calls.append((k, lo, hi))
        if hi <= lo:
            return
        ts = [t for name, t in grads.items() if t is not None and name in self.params]
        if ts and ts[0].is_cuda:
            dev = ts[0].device
            if self.comm is None:
                self.comm = torch.cuda.Stream(dev)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(ev)
                for t in ts:
                    self._works.append(dist.all_reduce(t[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                    t.record_stream(self.comm)
        else:
            for t in ts:
                dist.all_reduce(t[lo:hi], op=dist.ReduceOp.SUM, group=self.group)

    def __exit__(self, exc_type, exc, tb):
        from . import rasterizer as rz
        if self._active():
            rz.GRADIENT_RANGE_HOOK, rz.GRADIENT_RANGES = self._saved
        if exc_type is not None or self._tensors is None:
            for w in self._works:
                w.wait()
            return False
        cuda = any(t is not None and t.is_cuda for t in self._tensors.values())
        if cuda:
            dev = next(t.device for t in self._tensors.values() if t is not None)
            cur = torch.cuda.current_stream(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)                      # the backward pass's own kernels end here ...
            for w in self._works:
                w.wait()                        # (the current stream waits for the collective's stream)
            if self.comm is not None:
                cur.wait_stream(self.comm)
            e1.record(cur)                      # ... and here the sums are in: the gap is what the exchange was not hidden behind
            self._events = (e0, e1)
        with torch.no_grad():
            for name, p_ in self.params.items():
                t = self._tensors.get(name)
                if t is None or p_.grad is None:
                    continue
                if p_.grad.data_ptr() != t.data_ptr():   # autograd made its own copy (possibly before the sum was in): hand it the summed tensor
                    p_.grad = t.view_as(p_)
        return False

    def exposed_ms(self):
        """milliseconds between the end of the backward pass's kernels and the end of the exchange on the compute stream (GPU only;
        synchronises on the second event)"""
        if self._events is None:
            return None
        self._events[1].synchronize()
        return float(self._events[0].elapsed_time(self._events[1]))
