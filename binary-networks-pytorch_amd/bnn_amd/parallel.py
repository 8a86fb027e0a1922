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


def _host_staged(group) -> bool:
    """gloo moves host memory: device tensors are staged through the host around the collective.  Only tests and
    ``bench.py --backend gloo`` (several ranks on ONE GPU, which RCCL refuses) come this way; it synchronises the
    stream, which the RCCL path never does."""
    return dist.get_backend(group) == "gloo"


def all_gather_rows(y: torch.Tensor, group=None) -> torch.Tensor:
    """``[n, ...]`` per rank -> ``[world * n, ...]`` on every rank, rank r's rows in block r: ONE collective
    (``all_gather_into_tensor`` — RCCL ``ncclAllGather`` on the current stream)."""
    world = dist.get_world_size(group)
    if y.is_cuda and _host_staged(group):
        host = y.cpu()
        out = host.new_empty((world * host.shape[0],) + tuple(host.shape[1:]))
        dist.all_gather_into_tensor(out, host, group=group)
        return out.to(y.device)
    out = y.new_empty((world * y.shape[0],) + tuple(y.shape[1:]))
    dist.all_gather_into_tensor(out, y, group=group)
    return out


def all_gather_scalar(value, device, dtype=torch.float64, group=None) -> list:
    """One number per rank -> the list of all ranks' numbers (bench bookkeeping: step times, shard sizes)."""
    world = dist.get_world_size(group)
    dev = torch.device("cpu") if _host_staged(group) else device
    mine = torch.tensor([value], device=dev, dtype=dtype)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine, group=group)
    return [t.item() for t in every]


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
        self._pending = None      # forward_even(overlap=True): the previous step's collective (torch.distributed Work)
        self._out = None

    @torch.no_grad()
    def forward(self, x_local: torch.Tensor) -> torch.Tensor:
        y = self.model(x_local).contiguous()
        if not (dist.is_available() and dist.is_initialized()):
            return y
        world = dist.get_world_size(self.group)
        if world == 1:
            return y
        # ragged shards (batch not divisible by world) are padded to the largest shard
        sizes = [int(n) for n in all_gather_scalar(y.shape[0], y.device, torch.int64, self.group)]
        n_max = max(sizes)
        if y.shape[0] < n_max:
            y = torch.cat([y, y.new_zeros((n_max - y.shape[0],) + y.shape[1:])], 0)
        out = all_gather_rows(y, self.group)
        if all(s == n_max for s in sizes):
            return out
        return torch.cat([out[r * n_max: r * n_max + sizes[r]] for r in range(world)], 0)

    @torch.no_grad()
    def forward_even(self, x_local: torch.Tensor, overlap: bool = False) -> torch.Tensor:
        """Fast path when every rank holds the same number of images: exactly one collective.

        ``overlap=True`` (throughput loops: ``bench.py``): the all-gather is issued asynchronously — RCCL runs it on
        its own stream behind the kernels that produced the logits, and the calling stream goes straight on to its next
        batch instead of idling through a latency-bound 1 MB-per-rank collective.  The returned tensor is valid after
        ``wait()`` (or a device synchronisation); the NEXT ``forward_even`` of this object waits for the collective
        before the model may overwrite the buffers it reads.  It is this object's ONE gather buffer: the next
        ``forward_even(overlap=True)`` overwrites it — copy it out in stream order (``wait()``, then ``.clone()``) to
        keep a step's logits across steps or to read them on another stream."""
        if overlap and self._pending is not None:
            self._pending.wait()
            self._pending = None
        y = self.model(x_local).contiguous()
        if not (dist.is_available() and dist.is_initialized()) or \
                (dist.get_world_size(self.group) == 1 and not self.force_collective):
            return y
        if not overlap or (y.is_cuda and _host_staged(self.group)):
            return all_gather_rows(y, self.group)
        world = dist.get_world_size(self.group)
        shape = (world * y.shape[0],) + tuple(y.shape[1:])
        if self._out is None or tuple(self._out.shape) != shape or self._out.device != y.device:
            self._out = y.new_empty(shape)
        self._pending = dist.all_gather_into_tensor(self._out, y, group=self.group, async_op=True)
        return self._out

    def wait(self) -> None:
        """Make the current stream wait for the collective of the last ``forward_even(overlap=True)``."""
        if self._pending is not None:
            self._pending.wait()
            self._pending = None
