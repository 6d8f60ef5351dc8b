"""One process per GPU, batch sharded across ranks: every image renders and back-propagates
independently, so the render path itself needs NO collective.  What crosses xGMI in a training step is the gradient of
whatever produced the attributes (the encoders, which stay on stock PyTorch-ROCm): one averaged all-reduce per step.

xGMI is point-to-point (7 links x ~153 GB/s per GPU) and a ring all-reduce is bound by one link per direction, so small
messages are latency-bound and many small all-reduces serialise: gradients are flattened into a few LARGE buckets
(default 64 MiB) before `all_reduce`, rather than issuing one collective per parameter.

backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests (world_size 2).
"""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, world, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    device = torch.device("cuda", local) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = backend or ("nccl" if use_cuda else "gloo")
        if backend == "nccl":
            dist.init_process_group(backend, device_id=device)
        else:
            dist.init_process_group(backend)
    return rank, world, device


def shard_bounds(n, rank, world):
    """Contiguous, near-equal slice [lo, hi) of n units for `rank` (first n % world ranks get one extra)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard(batch, rank, world):
    """Slice every tensor of a dict (or a single tensor) along dim 0; non-tensors pass through."""
    if torch.is_tensor(batch):
        lo, hi = shard_bounds(batch.shape[0], rank, world)
        return batch[lo:hi]
    n = next(v.shape[0] for v in batch.values() if torch.is_tensor(v))
    lo, hi = shard_bounds(n, rank, world)
    return {k: (v[lo:hi] if torch.is_tensor(v) else v) for k, v in batch.items()}


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    """bench.py's timing rule: the slowest rank defines the step time."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allreduce_mean_(tensors, bucket_bytes=64 << 20):
    """In-place mean over ranks of a list of tensors, through few large flat buckets (see module docstring).
    With equal per-rank batch sizes, per-rank mean losses + this average = the global-batch mean gradient."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    world = dist.get_world_size()
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([t.reshape(-1) for t in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world)
        o = 0
        for t in bucket:
            t.copy_(flat[o:o + t.numel()].view_as(t))
            o += t.numel()
        bucket, size = [], 0

    for t in tensors:
        if t is None:
            continue
        nb = t.numel() * t.element_size()
        if bucket and (size + nb > bucket_bytes or t.dtype != bucket[0].dtype):
            flush()
        bucket.append(t)
        size += nb
    flush()


def broadcast_(tensor, src=0):
    """e.g. the EM template update (trainer.py:1100): rank 0's vertices_init to everyone."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(tensor, src=src)
    return tensor
