"""Data-parallel helpers: one process per GPU (``torch.distributed``; backend ``nccl`` is RCCL over xGMI on
ROCm, ``gloo`` in the CPU tests).

The fusion forward has no cross-sample coupling (per-token LayerNorm, per-row softmax, per-sample mean), so the
batch is sharded contiguously across ranks and the forward needs NO collective.  The only exchange step of the
path is the gradient average of a training step: ``allreduce_mean_`` does it with a few large
flat buckets (the xGMI mesh is point-to-point, 7 links per GPU: a handful of multi-MB messages beat 125 small ones).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise the default process group from torchrun's environment.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local_rank


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n samples for `rank`; the first n % world ranks take one extra sample."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(tensors: Sequence[Optional[torch.Tensor]], rank: int, world: int) -> List[Optional[torch.Tensor]]:
    """Slice every modality tensor (batch-first) to this rank's shard; ``None`` (missing modality) passes through."""
    n = next(t.shape[0] for t in tensors if t is not None)
    lo, hi = shard_bounds(n, rank, world)
    return [None if t is None else t[lo:hi] for t in tensors]


def gather_outputs(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """All-gather per-rank outputs of a sharded forward back into batch order (ragged shards allowed)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    pad = max(hi - lo for lo, hi in sizes)
    buf = local.new_zeros((pad,) + tuple(local.shape[1:]))
    buf[: local.shape[0]] = local
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    return torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=0)


def allreduce_mean_(tensors: Iterable[torch.Tensor], bucket_bytes: int = 32 << 20) -> None:
    """In-place average of a list of tensors (e.g. gradients) across ranks using flat buckets."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    world = dist.get_world_size()
    bucket: List[torch.Tensor] = []
    size = 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        if len(bucket) == 1 and bucket[0].is_contiguous():     # already flat (healnet_amd.train.FlatParameters.grads): in place
            dist.all_reduce(bucket[0], op=dist.ReduceOp.SUM)
            bucket[0].div_(world)
            bucket, size = [], 0
            return
        flat = torch.cat([t.reshape(-1) for t in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world)
        off = 0
        for t in bucket:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
        bucket, size = [], 0

    for t in tensors:
        nbytes = t.numel() * t.element_size()
        if bucket and (size + nbytes > bucket_bytes or t.dtype != bucket[0].dtype):
            flush()
        bucket.append(t)
        size += nbytes
    flush()


def max_over_ranks(value: float, device: Optional[torch.device] = None) -> float:
    """bench.py timing contract: the slowest rank defines the step time."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
