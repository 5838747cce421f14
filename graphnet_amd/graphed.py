"""hipGraph capture of a whole training step (forward, backward, gradient copy, Adam).

After the kernels were made fast the step became launch-bound on the host (~300 kernel launches
and ~100 ctypes calls per step); a captured graph replays the identical kernels with one launch.
The graph is keyed by the batch *shapes* (N pulses, B events): a batch of the same shape is copied
into the static input buffers and replayed, a new shape is captured once (pad batches to a few
shape buckets in real training).  The gradient all-reduce of the data-parallel path stays outside
the captured graphs (graph 1: zero-grad + forward + backward, eager: one flat RCCL all-reduce,
graph 2: optimizer step), so RCCL is never part of a capture.

Semantics are those of ``StandardModel.fit``'s eager loop (``models/easy_model.py:237-256``).

Status: opt-in (``bench.py --graph``); the replay follows the eager loop to ~1e-5 in the loss (not bit for bit: the
library GEMMs of the tiny read-out pick other algorithms under capture).  History: in round 1 replays 7..26 of a
B = 1024 step raised "Memory access fault ... write access to a read-only page".  The one thing that set the faulting
step apart from every clean one was a kernel with PRIVATE-SEGMENT (scratch) memory in the captured graph:
``edge_fwd_ws_kernel<22,21,8>`` spilled 48 bytes per lane.  Eager launches of that kernel never faulted; with the
spill removed (csrc/edgeconv_v2.hip: one select-free hot path) 38 replays of the same step, same size, same seeds run
clean and reproduce the eager losses.  Rule kept by ``tests/test_cabi_exports.py``: no kernel of the library may have
a non-zero private segment.
At least one eager warm-up step is required: the optimizer creates its state on the first ``step()``, and
state created *inside* the capture would be re-initialised by every replay.
"""
from __future__ import annotations

from typing import Any, Dict, Optional, Tuple

import torch
import torch.distributed as dist

from .data import Data
from .parallel import FlatGradAllReduce


class GraphedTrainStep:
    MAX_IN_FLIGHT = 2

    def __init__(self, model: torch.nn.Module, optimizer: torch.optim.Optimizer,
                 sync: Optional[FlatGradAllReduce] = None, scheduler: Any = None, warmup: int = 2):
        self.model, self.opt, self.sched = model, optimizer, scheduler
        self.sync = sync if sync is not None else FlatGradAllReduce(model.parameters())
        self.warmup = max(1, int(warmup))          # >= 1: optimizer state must exist before the capture
        self._graphs: Dict[Tuple, Tuple] = {}
        self._events: list = []
        self.distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        dev = self.sync.flat.device
        self._lr = []
        for group in optimizer.param_groups:
            if not group.get("capturable", False):
                raise ValueError("GraphedTrainStep needs an optimizer constructed with capturable=True")
            # the learning rate must live on the device to change between replays (LR schedules)
            t = torch.tensor(float(group["lr"]), dtype=torch.float32, device=dev)
            group["lr"] = t
            self._lr.append(t)

    @staticmethod
    def _key(batch: Data) -> Tuple:
        return tuple((k, tuple(v.shape), str(v.dtype)) for k, v in sorted(batch.items()) if isinstance(v, torch.Tensor))

    def _fwd_bwd(self, batch: Data) -> torch.Tensor:
        self.sync.zero_grad()
        loss = self.model.shared_step(batch)
        loss.backward()
        self.sync.gather_into_flat()
        return loss

    def _capture(self, batch: Data):
        static = Data(**{k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()})
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup):                      # eager warm-up on a side stream
                self._fwd_bwd(static)
                if self.distributed:
                    self.sync.all_reduce()
                self.opt.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1):
            loss = self._fwd_bwd(static)
            if not self.distributed:
                self.opt.step()
        g2 = None
        if self.distributed:
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, pool=g1.pool()):
                self.opt.step()
        return static, g1, g2, loss

    def __call__(self, batch: Data) -> torch.Tensor:
        key = self._key(batch)
        if key not in self._graphs:
            self._graphs[key] = self._capture(batch)
        static, g1, g2, loss = self._graphs[key]
        if batch is not static:
            for k, v in batch.items():
                if isinstance(v, torch.Tensor):
                    static[k].copy_(v, non_blocking=True)
        # At most MAX_IN_FLIGHT replays are queued (keeps the host from running arbitrarily far ahead).
        if len(self._events) >= self.MAX_IN_FLIGHT:
            self._events.pop(0).synchronize()
        g1.replay()
        if g2 is not None:
            self.sync.all_reduce()
            g2.replay()
        ev = torch.cuda.Event()
        ev.record()
        self._events.append(ev)
        if self.sched is not None:
            self.sched.step()                               # assigns python floats ...
            for group, t in zip(self.opt.param_groups, self._lr):
                if not (isinstance(group["lr"], torch.Tensor) and group["lr"] is t):
                    t.fill_(float(group["lr"]))             # ... moved into the captured tensor
                    group["lr"] = t
        return loss
