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


def allreduce_mean_(tensors, bucket_bytes=64 << 20, local_weight=None):
    """In-place mean over ranks of a list of tensors, through few large flat buckets (see module docstring).
    With equal per-rank batch sizes, per-rank mean losses + this average = the global-batch mean gradient.  When the shards are
    NOT equal (shard_bounds hands the first B % world ranks one extra image) pass ``local_weight`` = this rank's batch size: the
    result is then sum_r w_r * t_r / sum_r w_r, the global-batch mean of per-rank batch means."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    world = dist.get_world_size()
    scale_in, scale_out = 1.0, 1.0 / world
    if local_weight is not None:
        first = next((t for t in tensors if t is not None), None)
        wt = torch.tensor([float(local_weight)], dtype=torch.float64, device=first.device if first is not None else "cpu")
        dist.all_reduce(wt, op=dist.ReduceOp.SUM)
        scale_in, scale_out = float(local_weight), 1.0 / float(wt.item())
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([t.reshape(-1) for t in bucket])
        if scale_in != 1.0:
            flat.mul_(scale_in)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.mul_(scale_out)
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


class GradAllReducer:
    """Mean all-reduce of ONE flat gradient buffer per step, issued on a side stream so that it overlaps the next step's render
    kernels (the render path needs no collective, so nothing on the compute streams ever waits for xGMI except the optimizer).

    SURVEY 8(e): the message is the attribute-producing networks' gradient (135-200 MB fp32).  xGMI is point-to-point, a ring
    is bound by one link per direction, so the buffer goes out as few LARGE chunks (default 64 MiB), back to back on the
    communicator's stream.  Usage per step:  launch() after the backward that filled ``flat``;  wait() before ``flat`` is read
    (optimizer) or overwritten (next backward).  CPU / gloo: same calls, asynchronous work handles instead of streams."""

    def __init__(self, flat, chunk_bytes=64 << 20):
        self.flat = flat
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        n = max(1, chunk_bytes // flat.element_size())
        self.chunks = [flat[o:o + n] for o in range(0, flat.numel(), n)]
        self.cuda = flat.is_cuda
        self.side = torch.cuda.Stream(flat.device) if self.cuda and self.world > 1 else None
        self.pending = []
        self.launched = 0

    def launch(self):
        if self.world == 1:
            return
        self.wait()                                              # at most one reduction of this buffer in flight
        if self.cuda:
            self.side.wait_stream(torch.cuda.current_stream(self.flat.device))    # the gradients must have been produced
            with torch.cuda.stream(self.side):
                for c in self.chunks:
                    dist.all_reduce(c, op=dist.ReduceOp.SUM)     # enqueued after everything on `side`; the host does not block
                    c.mul_(1.0 / self.world)
        else:
            self.pending = [dist.all_reduce(c, op=dist.ReduceOp.SUM, async_op=True) for c in self.chunks]
        self.launched += 1

    def wait(self):
        """Make the current stream (CPU: the host) wait for the reduction in flight, if any."""
        if self.world == 1:
            return
        if self.cuda:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.side)
        else:
            for w in self.pending:
                w.wait()
            if self.pending:
                self.flat.mul_(1.0 / self.world)
            self.pending = []

    def bytes_per_step(self):
        return self.flat.numel() * self.flat.element_size()


class ReduceSchedule:
    """Which steps of a loop launch the gradient all-reduce.  Every rank must issue the SAME collectives in the same order, so the
    decision may only depend on things that are equal on all ranks: a step counter that is reset at points all ranks reach together, and
    a cadence K agreed on from max-over-ranks timings.  A loop that runs for a TIME (bench.py's settling phase) executes a different
    number of steps on every rank: the schedule must be off() there -- ranks that launch different numbers of all-reduces deadlock."""

    def __init__(self, reducer):
        self.reducer, self.every, self.k = reducer, None, 0

    def off(self):
        self.every = None

    def start(self, every):
        """Switch the cadence on (or restart it): the next step launches a reduction, then every `every`-th one."""
        self.every, self.k = max(1, int(every)), 0

    def step(self):
        """Call once per step, after the step's work has been enqueued."""
        if self.reducer is None or self.every is None:
            return False
        k = self.k
        self.k += 1
        if k % self.every == 0:
            self.reducer.launch()
            return True
        return False


def broadcast_(tensor, src=0):
    """e.g. the EM template update (trainer.py:1100): rank 0's vertices_init to everyone."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(tensor, src=src)
    return tensor


def ddp_step_check(net, forward_loss, device):
    """One data-parallel step of ``net`` the way a trainer built on this path takes it, checked against its definition:
      1. this rank's gradient of ``forward_loss(net)`` on its own shard, WITHOUT synchronisation (``no_sync``);
      2. the same step through ``torch.nn.parallel.DistributedDataParallel`` -- bucketed all-reduce (RCCL over xGMI on GPUs, gloo in the CPU
         tests) overlapped with the backward;
    and returns (ddp module, max |DDP gradient - mean over ranks of the local gradients|).  With equal per-rank batches the mean of the per-rank
    batch-mean gradients IS the global-batch gradient; the reference has nothing to compare with (trainer.py:94-95 is DataParallel)."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    world = dist.get_world_size() if dist.is_initialized() else 1
    ddp = DDP(net, device_ids=[device.index] if device.type == "cuda" else None, bucket_cap_mb=64, gradient_as_bucket_view=True) if world > 1 else net
    params = [p for p in net.parameters() if p.requires_grad]
    for p in params:
        p.grad = None
    if world > 1:
        with ddp.no_sync():
            forward_loss(ddp).backward()
    else:
        forward_loss(ddp).backward()
    local = [p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p) for p in params]
    allreduce_mean_(local)                                       # the definition: mean over ranks, through this module's flat buckets
    for p in params:
        p.grad = None
    forward_loss(ddp).backward()                                 # DDP's own bucketed, overlapped reduction
    err = 0.0
    for p, ref in zip(params, local):
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        err = max(err, float((g - ref).abs().max()))
    return ddp, err


def ddp_gradient_buffer(dr, B, H, W, device, rank, seed=0):
    """bench.py, N > 1: the REAL gradient message of a data-parallel trainer on this path.  Builds trainer_step.AttributeNet (two ResNet-18
    trunks + conv stacks on stock PyTorch: the networks that PRODUCE the render path's attributes), takes one DistributedDataParallel step of
    it through the render path on this rank's shard (ddp_step_check), and returns (flat fp32 buffer holding its gradient, info dict)."""
    import importlib
    ts = importlib.import_module(__package__ + ".trainer_step")
    syn = importlib.import_module(__package__ + ".synthetic")
    torch.manual_seed(seed)                                      # identical initial weights on every rank
    net = ts.AttributeNet(dr.vertices_init, bg=True).to(device)
    _, gt = syn.synthetic_batch(dr.vertices_init, B, H, W, seed=1000 + rank)     # this rank's shard of the global batch
    x = gt.to(device)

    def forward_loss(m):
        att = m(x)
        loss, _, _ = dr.render_recon(x, no_mask=True, **att)
        return loss

    ddp, err = ddp_step_check(net, forward_loss, device)
    flat = torch.cat([p.grad.detach().reshape(-1).float() for p in net.parameters() if p.grad is not None]).contiguous()
    info = {"module": "trainer_step.AttributeNet", "params": int(sum(p.numel() for p in net.parameters())),
            "gradient_mb": round(flat.numel() * 4 / 1e6, 1), "ddp_vs_mean_of_local_max_abs_err": err,
            "bucket_cap_mb": 64, "note": "one DistributedDataParallel step (bucketed all-reduce overlapped with the backward) checked against the mean over "
                                          "ranks of the unsynchronised local gradients; its gradient buffer is what the timed region keeps reducing"}
    del ddp
    return flat, info
