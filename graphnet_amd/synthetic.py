"""Synthetic IceCube-86 pulse batches (the workload of BASELINE.json configs[1]/[2]).

Specification: SURVEY.md §8(d) "Synthetic inputs".  Pulses per event follow a clipped
log-normal (mean ~150), DOMs are drawn around a random vertex from the IceCube-86 geometry
(5407 sensors; ``tests/golden/icecube86_geometry.npz``, extracted as data from the reference's
``data/geometry_tables/icecube/icecube86.parquet``), several pulses share one DOM (duplicate
xyz, as in real data), features are in ``FEATURES.ICECUBE86`` order
(``data/constants.py:7-15``) and standardised with ``IceCube86`` (``icecube.py:35-48``).
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch

from .data import Batch
from .detector import IceCube86

FEATURES_ICECUBE86 = ["dom_x", "dom_y", "dom_z", "dom_time", "charge", "rde", "pmt_area"]
_GEO = None


def _analytic_geometry() -> np.ndarray:
    """Fallback hexagonal 86-string layout (78 strings at 125 m, 8 dense in-fill strings)."""
    pts = []
    for q in range(-5, 6):
        for r in range(-5, 6):
            if abs(q + r) <= 5:
                pts.append((125.0 * (q + r / 2.0), 125.0 * r * np.sqrt(3) / 2.0))
    pts = sorted(pts, key=lambda p: p[0] ** 2 + p[1] ** 2)[:78]
    pts += [(40.0 * np.cos(a), 40.0 * np.sin(a)) for a in np.linspace(0, 2 * np.pi, 8, endpoint=False)]
    rows = []
    for s, (x, y) in enumerate(pts):
        dense = s >= 78
        zs = np.linspace(-500, -160, 60) if dense else np.linspace(-500, 500, 60)
        for z in zs:
            rows.append((x, y, z, 1.35 if dense else 1.0, 0.0444))
    return np.asarray(rows, dtype=np.float32)


def icecube86_geometry() -> np.ndarray:
    """``[n_sensors, 5]`` float32: dom_x, dom_y, dom_z, rde, pmt_area."""
    global _GEO
    if _GEO is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                            "tests", "golden", "icecube86_geometry.npz")
        geo = np.load(path)["table"] if os.path.exists(path) else _analytic_geometry()
        # in-ice sensors only: the table also lists 324 IceTop tanks (z ~ +1950 m) and 3 rows
        # without a relative efficiency (NaN), which real in-ice pulse series never contain
        keep = np.isfinite(geo).all(1) & (geo[:, 2] < 600.0)
        _GEO = np.ascontiguousarray(geo[keep])
    return _GEO


def synthetic_icecube86_raw(n_events: int, seed: int = 20241016, mean_scale: float = 130.0,
                            n_min: int = 8, n_max: int = 2000, count_range: Optional[tuple] = None):
    """Raw (un-standardised) pulses: ``x[N,7]`` float32, ``ptr[B+1]`` int64, ``energy[B]``.
    ``count_range=(lo, hi)``: pulses per event log-uniform in [lo, hi] instead of the clipped log-normal
    (BASELINE configs[3]: "mixed 50-3000 pulses/event")."""
    rng = np.random.default_rng(seed)
    geo = icecube86_geometry()
    if count_range is not None:
        lo_c, hi_c = count_range
        n = np.rint(np.exp(rng.uniform(np.log(lo_c), np.log(hi_c), n_events))).astype(np.int64)
    else:
        n = np.clip(np.rint(rng.lognormal(np.log(mean_scale), 0.55, n_events)), n_min, n_max).astype(np.int64)
    ptr = np.zeros(n_events + 1, np.int64)
    ptr[1:] = np.cumsum(n)
    x = np.empty((int(ptr[-1]), 7), np.float32)
    lo, hi = geo[:, :3].min(0), geo[:, :3].max(0)
    for b in range(n_events):
        ni = int(n[b])
        m = min(int(np.ceil(ni / 1.4)), geo.shape[0])
        vertex = rng.uniform(lo * 0.8, hi * 0.8)
        d2 = ((geo[:, :3] - vertex) ** 2).sum(1)
        keys = -d2 / (2.0 * 150.0 ** 2) + rng.gumbel(size=geo.shape[0])
        doms = np.argpartition(-keys, m - 1)[:m]
        mult = rng.geometric(0.7, m)                  # = 1 + Geometric(0.7) on {0,1,..}
        ids = np.repeat(doms, mult)
        if len(ids) < ni:
            ids = np.concatenate([ids, rng.choice(doms, ni - len(ids))])
        ids = ids[:ni]
        dist = np.sqrt(d2[ids])
        t = 1.0e4 + dist / 0.3 + rng.exponential(200.0, ni)
        order = np.argsort(t, kind="stable")
        ids, t = ids[order], t[order]
        q = np.maximum(rng.lognormal(0.0, 0.7, ni), 0.05)
        s = slice(int(ptr[b]), int(ptr[b + 1]))
        x[s, 0:3] = geo[ids, 0:3]
        x[s, 3] = t
        x[s, 4] = q
        x[s, 5] = geo[ids, 3]
        x[s, 6] = geo[ids, 4]
    energy = (10.0 ** rng.uniform(0.0, 4.0, n_events)).astype(np.float32)
    return x, ptr, energy


def synthetic_icecube86_batch(n_events: int, seed: int = 20241016, device: Optional[str] = None,
                              **kwargs) -> Batch:
    """Standardised batch in batched-CSR form (no edges: the backend builds layer-1 k-NN)."""
    x, ptr, energy = synthetic_icecube86_raw(n_events, seed, **kwargs)
    xt = IceCube86()(torch.from_numpy(x), FEATURES_ICECUBE86)
    ptr_t = torch.from_numpy(ptr)
    n_pulses = (ptr_t[1:] - ptr_t[:-1]).to(torch.int32)
    b = Batch(x=xt)
    b.ptr = ptr_t
    b.batch = torch.repeat_interleave(torch.arange(n_events, dtype=torch.int64), n_pulses.long())
    b.n_pulses = n_pulses
    b.energy = torch.from_numpy(energy)
    if device is not None:
        b.to(device)
    return b


def synthetic_track_batch(n_events: int, seed: int = 7, mean_pulses: float = 1.0e4,
                          device: Optional[str] = None) -> Batch:
    """Config-5 style high-energy tracks: ~1e4 pulses on DOMs within 120 m of a random line."""
    rng = np.random.default_rng(seed)
    geo = icecube86_geometry()
    xs, ptr = [], [0]
    for _ in range(n_events):
        ni = int(max(4000, rng.normal(mean_pulses, 0.15 * mean_pulses)))
        p0 = rng.uniform(-300, 300, 3)
        dirv = rng.normal(size=3)
        dirv /= np.linalg.norm(dirv)
        rel = geo[:, :3] - p0
        perp = np.linalg.norm(rel - np.outer(rel @ dirv, dirv), axis=1)
        doms = np.nonzero(perp < 120.0)[0]
        if len(doms) < 16:
            doms = np.argsort(perp)[:64]
        ids = rng.choice(doms, ni)
        t = 1.0e4 + (rel[ids] @ dirv) / 0.3 + rng.exponential(300.0, ni)
        order = np.argsort(t, kind="stable")
        ids, t = ids[order], t[order]
        a = np.empty((ni, 7), np.float32)
        a[:, 0:3] = geo[ids, 0:3]
        a[:, 3] = t
        a[:, 4] = np.maximum(rng.lognormal(0.3, 0.9, ni), 0.05)
        a[:, 5] = geo[ids, 3]
        a[:, 6] = geo[ids, 4]
        xs.append(a)
        ptr.append(ptr[-1] + ni)
    x = IceCube86()(torch.from_numpy(np.concatenate(xs)), FEATURES_ICECUBE86)
    ptr_t = torch.tensor(ptr, dtype=torch.int64)
    n_pulses = (ptr_t[1:] - ptr_t[:-1]).to(torch.int32)
    b = Batch(x=x)
    b.ptr = ptr_t
    b.batch = torch.repeat_interleave(torch.arange(n_events, dtype=torch.int64), n_pulses.long())
    b.n_pulses = n_pulses
    b.energy = torch.from_numpy((10.0 ** rng.uniform(3.0, 6.0, n_events)).astype(np.float32))
    if device is not None:
        b.to(device)
    return b
