"""Data-parallel collocation batches: one process per GPU, RCCL over xGMI via `torch.distributed`.

New functionality (the reference is single-process, SURVEY.md §2 "Parallelism inventory: none").
Every collocation point is independent, so a batch is split into `world` contiguous row ranges and
the only exchange is ONE sum all-reduce per step of `[flat gradient || loss scalars]`
(numel(theta) + a few floats: 0.17-3.2 MB, latency-bound on xGMI — so it is a single fused
message, not per-tensor buckets).  Parameters stay replicated and bit-identical across ranks
because every rank applies the same reduced gradient.
"""

from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

Tensor = torch.Tensor


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Rows [lo, hi) of rank `rank`: contiguous, sizes differ by at most one, union = [0, n)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_points(x: Tensor, t: Tensor, group=None) -> Tuple[Tensor, Tensor, int]:
    """This rank's rows of an identically-sampled global batch, and the global row count."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(x.shape[0], rank, world)
    return x[lo:hi].contiguous(), t[lo:hi].contiguous(), x.shape[0]


def sharded_compute_loss(pde, model, x: Tensor, t: Tensor, group=None) -> Dict[str, Tensor]:
    """`pde.compute_loss` on this rank's shard; `total` is the LOCAL objective whose gradients SUM to the global one.

    `losses["residual"]` is the local partial sum divided by the global N (sum over ranks = global mean);
    call `all_reduce_gradients(..., scalars=[losses["residual"]])` after backward to obtain the global value.
    """
    world = dist.get_world_size(group)
    xs, ts, n_total = shard_points(x, t, group)
    return pde.compute_loss(model, xs, ts, n_total=n_total, aux_scale=1.0 / world)


def all_reduce_gradients(params: Sequence[Tensor], group=None, scalars: Optional[List[Tensor]] = None) -> Optional[Tensor]:
    """ONE sum all-reduce of every `.grad` plus optional scalar tensors; returns the reduced scalars."""
    ps = [p for p in params if p.grad is not None]
    if not ps and not scalars:
        return None
    dev = ps[0].grad.device if ps else scalars[0].device
    parts = [p.grad.reshape(-1) for p in ps]
    ns = len(scalars) if scalars else 0
    if ns:
        parts.append(torch.stack([s.detach().reshape(()).to(dev) for s in scalars]))
    flat = torch.cat(parts)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for p in ps:
        n = p.grad.numel()
        p.grad.copy_(flat[off : off + n].view_as(p.grad))
        off += n
    return flat[off : off + ns] if ns else None


def broadcast_parameters(module: torch.nn.Module, group=None, src: int = 0) -> None:
    """Make replicas identical before the first step (theta_0 built from different RNG states otherwise)."""
    with torch.no_grad():
        tensors = list(module.state_dict().values())
        if not tensors:
            return
        flat = torch.cat([v.reshape(-1).float() for v in tensors])
        dist.broadcast(flat, src=src, group=group)
        off = 0
        for v in tensors:
            n = v.numel()
            v.copy_(flat[off : off + n].view_as(v))
            off += n
