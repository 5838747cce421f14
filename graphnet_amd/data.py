"""Minimal ``Data`` / ``Batch`` containers with the attribute layout the reference's path uses.

The reference passes ``torch_geometric.data.Data`` / ``Batch`` objects between the data
loader and the model (``data/dataloader.py:12-18``, ``models/standard_model.py:96-109``).
torch_geometric is not a dependency here: these duck types carry the same attributes
(``x, edge_index, batch, ptr, n_pulses, num_graphs`` + arbitrary labels) and any object with
those attributes — including a real PyG ``Batch`` — is accepted by the backbone.
"""
from __future__ import annotations

from typing import Any, Iterator, List, Optional

import torch
from torch import Tensor


class Data:
    """Attribute bag for one event graph (mirrors the subset of PyG ``Data`` on the path)."""

    def __init__(self, x: Optional[Tensor] = None, edge_index: Optional[Tensor] = None, **kwargs: Any):
        self.__dict__["_store"] = {}
        self.x = x
        self.edge_index = edge_index
        for k, v in kwargs.items():
            setattr(self, k, v)

    def __getattr__(self, key: str) -> Any:
        store = self.__dict__["_store"]
        if key in store:
            return store[key]
        raise AttributeError(key)

    def __setattr__(self, key: str, value: Any) -> None:
        self.__dict__["_store"][key] = value

    def __getitem__(self, key: str) -> Any:
        return self.__dict__["_store"][key]

    def __setitem__(self, key: str, value: Any) -> None:
        self.__dict__["_store"][key] = value

    def __contains__(self, key: str) -> bool:
        return key in self.__dict__["_store"]

    def keys(self) -> List[str]:
        return [k for k, v in self.__dict__["_store"].items() if v is not None]

    def items(self) -> Iterator:
        return ((k, self[k]) for k in self.keys())

    @property
    def num_nodes(self) -> int:
        return 0 if self.x is None else int(self.x.shape[0])

    def to(self, device: Any, non_blocking: bool = False) -> "Data":
        for k, v in list(self.__dict__["_store"].items()):
            if isinstance(v, Tensor):
                self.__dict__["_store"][k] = v.to(device, non_blocking=non_blocking)
        return self

    def __repr__(self) -> str:
        parts = []
        for k, v in self.items():
            parts.append(f"{k}={list(v.shape)}" if isinstance(v, Tensor) else f"{k}={v!r}")
        return f"{self.__class__.__name__}({', '.join(parts)})"


class Batch(Data):
    """Concatenation of event graphs in batched-CSR form (``x`` rows grouped by event,
    ``ptr[B+1]`` offsets, ``batch[N]`` event ids), as ``Batch.from_data_list`` produces."""

    @property
    def num_graphs(self) -> int:
        if "ptr" in self and self.ptr is not None:
            return int(self.ptr.shape[0]) - 1
        return int(self.n_pulses.shape[0])

    @classmethod
    def from_data_list(cls, graphs: List[Data]) -> "Batch":
        """Restates PyG's concatenation rules for the attributes on this path:
        tensors with at least one dimension are concatenated along dim 0, ``edge_index`` along dim 1 after
        being offset by the cumulative node count, 0-d tensors / Python scalars are stacked to ``[B]``."""
        out = cls()
        sizes = [g.num_nodes for g in graphs]
        ptr = torch.zeros(len(graphs) + 1, dtype=torch.int64)
        if sizes:
            ptr[1:] = torch.cumsum(torch.tensor(sizes, dtype=torch.int64), 0)
        keys = graphs[0].keys() if graphs else []
        for k in keys:
            vals = [g[k] for g in graphs]
            v0 = vals[0]
            if k == "edge_index":
                out[k] = torch.cat([v + int(ptr[i]) for i, v in enumerate(vals)], dim=1)
            elif isinstance(v0, Tensor) and v0.dim() >= 1:
                out[k] = torch.cat(vals, dim=0)           # node-level rows and per-event rows alike (e.g. a [1, 1] loss weight -> [B, 1])
            elif isinstance(v0, Tensor):
                out[k] = torch.stack(vals, dim=0)         # 0-d: n_pulses, scalar labels -> [B]
            elif isinstance(v0, (int, float)):
                out[k] = torch.tensor(vals)
            else:
                out[k] = vals
        out.ptr = ptr
        out.batch = torch.repeat_interleave(torch.arange(len(graphs), dtype=torch.int64),
                                            torch.tensor(sizes, dtype=torch.int64))
        return out


def select_events(batch: Batch, events) -> Batch:
    """The sub-batch holding ``events`` (ascending event ids of ``batch``) - what a ``DistributedSampler`` shard of the
    same events would have collated: node rows of the kept events, per-event rows, ``edge_index`` relabelled,
    ``ptr`` / ``batch`` rebuilt.  Works on any device; index arithmetic only."""
    ptr = batch.ptr.to(torch.int64)
    dev = ptr.device
    ev = torch.as_tensor(list(events) if not isinstance(events, Tensor) else events, dtype=torch.int64, device=dev)
    B, N = int(ptr.shape[0]) - 1, int(ptr[-1])
    sizes = (ptr[1:] - ptr[:-1])[ev]
    new_ptr = torch.zeros(ev.numel() + 1, dtype=torch.int64, device=dev)
    new_ptr[1:] = torch.cumsum(sizes, 0)
    n_new = int(new_ptr[-1])
    new_batch = torch.repeat_interleave(torch.arange(ev.numel(), dtype=torch.int64, device=dev), sizes)
    rows = torch.arange(n_new, dtype=torch.int64, device=dev) - new_ptr[new_batch] + ptr[ev][new_batch]
    out = Batch()
    for k, v in batch.items():
        if k in ("ptr", "batch"):
            continue
        if k == "edge_index":
            new_id = torch.full((N,), -1, dtype=torch.int64, device=dev)
            new_id[rows] = torch.arange(n_new, dtype=torch.int64, device=dev)
            keep = new_id[v[1].to(dev)] >= 0                      # edges never leave an event: the target decides
            out[k] = torch.stack([new_id[v[0].to(dev)[keep]], new_id[v[1].to(dev)[keep]]])
        elif isinstance(v, Tensor) and v.dim() >= 1 and int(v.shape[0]) == N and N != B:
            out[k] = v[rows.to(v.device)]
        elif isinstance(v, Tensor) and v.dim() >= 1 and int(v.shape[0]) == B:
            out[k] = v[ev.to(v.device)]
        elif isinstance(v, list) and len(v) == B:
            out[k] = [v[int(i)] for i in ev]
        else:
            out[k] = v
    out.ptr, out.batch = new_ptr, new_batch
    return out


def collate_fn(graphs: List[Data]) -> Batch:
    """``data/dataloader.py:12-18``: drop events with <= 1 pulse, then batch."""
    graphs = [g for g in graphs if int(g.n_pulses) > 1]
    return Batch.from_data_list(graphs)


class collator_sequence_buckleting:
    """``training/utils.py:31-67`` (the reference's spelling): events sorted by pulse count and cut at the given
    fractions into several ``Batch`` objects, so that a dense-padded attention never pads short events to the
    longest one.  ``StandardModel.forward`` takes the list as it is (``standard_model.py:96-108``)."""

    def __init__(self, batch_splits: List[float] = [0.8]):
        self.batch_splits = list(batch_splits)

    def __call__(self, graphs: List[Data]) -> List[Batch]:
        graphs = sorted((g for g in graphs if int(g.n_pulses) > 1), key=lambda g: int(g.n_pulses))
        cuts = [0.0] + self.batch_splits + [1.0]
        out = []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            part = graphs[int(lo * len(graphs)): int(hi * len(graphs))]
            if part:
                out.append(Batch.from_data_list(part))
        return out
