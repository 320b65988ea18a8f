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
    if world > 1 and not dist.is_initialized():
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
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


def allreduce_gradients(params, bucket_bytes=256 << 20):
    """Sum the per-view gradients of the replicated parameters over all ranks, in large flat buckets
    (few, big collectives: xGMI rings are per-link bound, so bucket sizes are hundreds of MB)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    bucket, size = [], 0
    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        off = 0
        for g in bucket:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n
        bucket, size = [], 0
    for g in grads:
        nbytes = g.numel() * g.element_size()
        if size + nbytes > bucket_bytes and bucket:
            flush()
        bucket.append(g)
        size += nbytes
    flush()
