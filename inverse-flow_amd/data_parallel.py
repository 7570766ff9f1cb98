"""Batch-sharded data parallelism for the inverse-conv layers: one process per GPU, one flat
gradient all-reduce per step over RCCL/xGMI (backend "nccl" is RCCL on ROCm; "gloo" on CPU for
the tests).

Replaces the reference's single-process torch.nn.DataParallel (scatter / replicate / gather and an
implicit NCCL broadcast+reduce every step: inf/if_multiGPU_imagenet32.py:410-411,
inf/train/experiment.py:162-165).  Batch elements are independent in all four ops of the path;
only dW (and other parameter gradients) is a sum over the batch, so the data path
needs exactly one collective: sum the flat gradient buffer, divide by the world size (the loss is
a per-rank mean, inf/train/experiment.py:192).
"""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment (1 process -> 0,0,1)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend=None):
    """Join the process group described by RANK/WORLD_SIZE/MASTER_ADDR/MASTER_PORT (no-op for 1 rank)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    return rank, local_rank, world


def shard_bounds(n, rank, world):
    """Contiguous [lo, hi) slice of a batch of n for this rank; the remainder goes to the first ranks."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(x, rank=None, world=None):
    if rank is None:
        rank, _, world = env_world()
    lo, hi = shard_bounds(x.shape[0], rank, world)
    return x[lo:hi]


class GradBucket:
    """One contiguous fp32 buffer holding every parameter gradient; parameters' .grad become views of
    it, so the step's collective is a single all-reduce launched right after the backward."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off, self.views = 0, []
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            p.grad = self.views[-1]
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def release(self):
        """detach the parameters from the bucket for one backward pass: their gradients arrive in fresh tensors"""
        for p in self.params:
            p.grad = None

    def gather(self):
        """the fresh gradients into the bucket with one multi-tensor copy (parameters that received none: zeros), and the
        parameters' .grad back onto their bucket views"""
        got = [(v, p.grad) for v, p in zip(self.views, self.params) if p.grad is not None]
        if len(got) != len(self.params):
            self.flat.zero_()
        if got:
            torch._foreach_copy_([v for v, _ in got], [g for _, g in got])
        for v, p in zip(self.views, self.params):
            p.grad = v

    def allreduce_mean(self, async_op=False, force=False):
        """Sum over ranks, divide by the world size.  Returns the work handle when async_op.  force: issue the collective in
        a process group of one rank too (a no-op in value: what a test of the captured step needs)."""
        if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
            return None
        world = dist.get_world_size()
        if async_op:
            work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=True)
            return _Scaled(work, self.flat, 1.0 / world)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.mul_(1.0 / world)
        return None


class _Scaled:
    def __init__(self, work, flat, scale):
        self.work, self.flat, self.scale = work, flat, scale

    def wait(self):
        self.work.wait()
        self.flat.mul_(self.scale)


def allreduce_mean_(tensor):
    """In-place mean over ranks of one gradient tensor (the single-layer bench's collective)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
        tensor.mul_(1.0 / dist.get_world_size())
    return tensor


def broadcast_parameters(module, src=0):
    """Identical weights on every rank (the reference re-broadcasts every step inside DataParallel)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src)
