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
    """-> (rank, world_size, local_rank). Reads the torchrun environment; no-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available() and local_rank >= torch.cuda.device_count():
        # more ranks than GPUs on this node (a test box): the ranks share devices, and RCCL refuses two ranks on one GPU
        local_rank %= torch.cuda.device_count()
        shared = True
    else:
        shared = False
    if world > 1 and not dist.is_initialized():
        if torch.cuda.is_available() and not shared and int(os.environ.get("LOCAL_WORLD_SIZE", "1")) <= torch.cuda.device_count():
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            if torch.cuda.is_available():
                torch.cuda.set_device(local_rank)
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


def allreduce_gradients(params, visible=None, sparse_below=0.4):
    """Sum the per-view gradients of the replicated parameters over all ranks. -> dict with what was exchanged.

    dense (default): every gradient tensor is all-reduced IN PLACE, all collectives in flight at once (no flat copy: the
    previous torch.cat + copy-back cost two extra passes over 1.42 GB at 6 M Gaussians; RCCL chunks a 1.15 GB tensor
    by itself, and xGMI rings are per-link bound, so few large collectives are the right shape).
    row-sparse (`visible` = this rank's bool [P] mask of the Gaussians its view touched, i.e. radii > 0): a Gaussian no
    view sees has an all-zero gradient row on every rank, so only the rows of the UNION of the masks are exchanged --
    one 1-byte-per-Gaussian MAX all-reduce for the union, a gather of those rows from every [P, ...] gradient into one
    [U, row] buffer, one all-reduce, a scatter back. Used when the union covers less than `sparse_below` of the
    Gaussians (two or three views of a large scene; eight views on a ring see most of it, and dense wins)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return dict(mode="none", bytes=0)
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return dict(mode="none", bytes=0)
    P = grads[0].shape[0]
    if visible is not None and all(g.shape[0] == P for g in grads):
        union = visible.to(torch.uint8)
        dist.all_reduce(union, op=dist.ReduceOp.MAX)
        idx = torch.nonzero(union, as_tuple=False).squeeze(1)
        U = int(idx.numel())
        if U < sparse_below * P:
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
    works = [dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True) for g in grads]
    for w in works:
        w.wait()
    return dict(mode="dense", bytes=sum(g.numel() * g.element_size() for g in grads))
