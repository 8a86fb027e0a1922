"""Batch-sharded multi-GPU inference: one process per GPU, one all-gather of logits.

The path shards by independent units (images; eval-mode BN has no cross-image state), so there is
no data-path collective except the final ``all_gather_into_tensor`` of the ``[B/G, classes]``
logits (RCCL over xGMI on MI355X; ``backend='nccl'`` is RCCL on ROCm).  At 1 MB per rank the
gather is latency-bound, so it is issued once per batch, not per micro-batch.

Reference analogue: ``nn.DataParallel`` scatter/gather in ``examples/cifar10.py:74-77`` and the DDP
evaluation loop of ``examples/imagenet.py:387-428``.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split of ``total`` items: the first ``total % world`` ranks get one more."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(x: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    lo, hi = shard_bounds(x.shape[0], rank, world)
    return x[lo:hi]


class ShardedInference(nn.Module):
    """Wraps a (replicated) model: ``forward(local_batch)`` returns the logits of ALL ranks."""

    def __init__(self, model: nn.Module, group: Optional[dist.ProcessGroup] = None,
                 force_collective: bool = False) -> None:
        super().__init__()
        self.model = model
        self.group = group
        # issue the all-gather even in a one-rank group (tests: the RCCL call, its stream ordering and its
        # behaviour next to graph replays are then exercised on a single GPU)
        self.force_collective = force_collective

    @torch.no_grad()
    def forward(self, x_local: torch.Tensor) -> torch.Tensor:
        y = self.model(x_local).contiguous()
        if not (dist.is_available() and dist.is_initialized()):
            return y
        world = dist.get_world_size(self.group)
        if world == 1:
            return y
        sizes = [None] * world
        # ragged shards (batch not divisible by world) are padded to the largest shard
        n_local = torch.tensor([y.shape[0]], device=y.device, dtype=torch.int64)
        all_n = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(all_n, n_local, group=self.group)
        sizes = [int(t.item()) for t in all_n]
        n_max = max(sizes)
        if y.shape[0] < n_max:
            y = torch.cat([y, y.new_zeros((n_max - y.shape[0],) + y.shape[1:])], 0)
        out = y.new_empty((world * n_max,) + y.shape[1:])
        dist.all_gather_into_tensor(out, y, group=self.group)
        if all(s == n_max for s in sizes):
            return out
        return torch.cat([out[r * n_max: r * n_max + sizes[r]] for r in range(world)], 0)

    @torch.no_grad()
    def forward_even(self, x_local: torch.Tensor) -> torch.Tensor:
        """Fast path when every rank holds the same number of images: exactly one collective."""
        y = self.model(x_local).contiguous()
        if not (dist.is_available() and dist.is_initialized()) or \
                (dist.get_world_size(self.group) == 1 and not self.force_collective):
            return y
        out = y.new_empty((dist.get_world_size(self.group) * y.shape[0],) + y.shape[1:])
        dist.all_gather_into_tensor(out, y, group=self.group)
        return out
