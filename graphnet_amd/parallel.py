"""Event-batch data parallelism: one process per GPU, one flat gradient all-reduce per step.

The reference delegates this to Lightning ``Trainer(strategy="ddp")``
(``models/easy_model.py:83-112``): event batches are sharded across ranks, the model is
replicated, gradients are summed with NCCL.  Here: the model is 5.5 MB, so
the exchange is latency-bound — all parameter gradients live in ONE flat fp32 buffer (the
``.grad`` tensors are views into it) and a single RCCL all-reduce over xGMI moves it after the
last backward kernel.  Works unchanged with the ``gloo`` backend on CPU tensors (tests).
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


class FlatGradAllReduce:
    """Keeps every parameter's ``.grad`` as a view of one flat buffer and all-reduces it."""

    def __init__(self, params: Iterable[torch.nn.Parameter], average: bool = True, group=None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.average = average
        self.group = group
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        self.views = []
        for p in self.params:
            v = self.flat[off: off + p.numel()].view_as(p)
            p.grad = v
            self.views.append(v)
            off += p.numel()

    def zero_grad(self) -> None:
        """Use instead of ``optimizer.zero_grad()``: ``.grad`` is dropped, so that autograd hands over each
        freshly computed gradient tensor as it is (no ``grad += new`` kernel per parameter); :meth:`gather_into_flat`
        then moves all of them into the flat buffer with one multi-tensor copy."""
        for p in self.params:
            p.grad = None

    def gather_into_flat(self) -> None:
        """Copy every parameter's gradient into its view of the flat buffer (ONE ``_foreach_copy_`` launch group
        instead of a kernel per parameter) and make the views the ``.grad`` tensors the optimizer steps on."""
        dst, src = [], []
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(p.grad.reshape(v.shape) if p.grad.shape != v.shape else p.grad)
            p.grad = v
        if dst:
            torch._foreach_copy_(dst, src)

    def all_reduce(self) -> None:
        """ONE sum-all-reduce of the flat gradient (RCCL over xGMI / gloo on CPU)."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            if self.average:
                self.flat.div_(dist.get_world_size(self.group))

    def __call__(self) -> None:
        self.gather_into_flat()
        self.all_reduce()


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Identical initial weights on every rank (what DDP does at construction)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def shard_events_by_pulses(n_pulses: Sequence[int], world_size: int) -> List[List[int]]:
    """Greedy longest-first assignment of events to ranks balancing total pulses (cost ~ N),
    in the spirit of ``data/dataset/samplers.py:160-292`` (length-matched batches)."""
    order = sorted(range(len(n_pulses)), key=lambda i: -int(n_pulses[i]))
    loads = [0] * world_size
    shards: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda j: loads[j])
        shards[r].append(i)
        loads[r] += int(n_pulses[i])
    for s in shards:
        s.sort()
    return shards


def shard_batch_by_pulses(batch, rank: Optional[int] = None, world_size: Optional[int] = None):
    """This rank's share of a GLOBAL batch that every rank holds (same loader, same order on all ranks): events are
    dealt to ranks by :func:`shard_events_by_pulses` (deterministic, so the shards are disjoint and cover the batch
    without any communication), balancing the pulses per rank because the cost of a step is ~ pulses, not events
    (the reference's length-matched sampler, ``data/dataset/samplers.py:160-292``)."""
    from .data import select_events
    if rank is None:
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if world_size == 1:
        return batch
    n = batch.n_pulses.detach().cpu().tolist()
    return select_events(batch, shard_events_by_pulses(n, world_size)[rank])


def check_equal_steps(n_steps: int, group=None) -> None:
    """Every step ends in a collective: ranks whose loaders disagree in length would hang in it.  One MIN/MAX
    all-reduce up front turns that into an error."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    t = torch.tensor([n_steps, -n_steps], dtype=torch.int64)
    backend = dist.get_backend(group)
    if backend == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    lo, hi = int(t[0]), -int(t[1])
    if lo != hi:
        raise RuntimeError(f"ranks disagree on the number of steps per epoch ({lo} .. {hi}): the gradient "
                           f"all-reduce would hang; give every rank the same number of batches")
