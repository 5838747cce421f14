"""oracle/detector_oracle.py — CPU restatement of the reference's per-detector feature standardisation.

TEST INFRASTRUCTURE ONLY (same rule as ``dynedge_oracle.py``: imported by ``tests/``, ``tests/golden/make_fixtures.py``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg; never by ``graphnet_amd/``).

Follows ``models/detector/detector.py:64-77`` (``Detector._standardize``: column ``idx`` of the fp32 feature tensor is
replaced by ``feature_map()[name](column)``, a ``KeyError`` for a name without a function) with the functions of
``models/detector/icecube.py:21-48`` (IceCube86), ``:84-113`` (IceCubeDeepCore), ``:116-170`` (IceCubeUpgrade) and
``models/detector/prometheus.py:11-39,365`` (ORCA150SuperDense = Prometheus), written out as the same Python
expressions on torch fp32 tensors, so every operation, its order and the Python-float constants are the reference's
(``x / 500.0`` is a division, ``(x - 1.0e04) / 3.0e4`` a subtraction then a division, ...).

Deliberately NOT built on ``graphnet_amd.detector`` (the product's table of op programs): this module is what the
product's host expressions AND the device kernel ``gn_standardize`` are checked against, and what generates the
``*_xstd`` inputs of ``tests/golden/oracle_expected.npz``.

PARITY: the arithmetic is elementwise IEEE fp32 (+ ``torch.log10``), which the reference evaluates with the same torch
functions; no reference-held output exists for it, so it is pinned only by construction.
"""
from __future__ import annotations

from typing import Callable, Dict, List

import torch
from torch import Tensor

Fn = Callable[[Tensor], Tensor]


def _identity(x: Tensor) -> Tensor:          # detector.py:79-81
    return x


def icecube86() -> Dict[str, Fn]:            # icecube.py:21-48
    return {
        "dom_x": lambda x: x / 500.0,
        "dom_y": lambda x: x / 500.0,
        "dom_z": lambda x: x / 500.0,
        "dom_time": lambda x: (x - 1.0e04) / 3.0e4,
        "charge": lambda x: torch.log10(x),
        "rde": lambda x: (x - 1.25) / 0.25,
        "pmt_area": lambda x: x / 0.05,
        "hlc": _identity,
    }


def icecube_deepcore() -> Dict[str, Fn]:     # icecube.py:84-113
    return {
        "dom_x": lambda x: x / 100.0,
        "dom_y": lambda x: x / 100.0,
        "dom_z": lambda x: (x + 350.0) / 100.0,
        "dom_time": lambda x: ((x / 1.05e04) - 1.0) * 20.0,
        "charge": _identity,
        "rde": lambda x: (x - 1.25) / 0.25,
        "pmt_area": lambda x: x / 0.05,
        "hlc": _identity,
    }


def icecube_upgrade() -> Dict[str, Fn]:      # icecube.py:116-170
    return {
        "dom_x": lambda x: x / 500.0,
        "dom_y": lambda x: x / 500.0,
        "dom_z": lambda x: x / 500.0,
        "dom_time": lambda x: (x / 2e04) - 1.0,
        "charge": lambda x: torch.log10(x) / 2.0,
        "rde": _identity,
        "pmt_area": lambda x: x / 0.05,
        "string": lambda x: (x - 50.0) / 50.0,
        "pmt_number": lambda x: x / 20.0,
        "dom_number": lambda x: (x - 60.0) / 60.0,
        "pmt_dir_x": _identity,
        "pmt_dir_y": _identity,
        "pmt_dir_z": _identity,
        "dom_type": lambda x: x / 130.0,
        "hlc": _identity,
    }


def prometheus() -> Dict[str, Fn]:           # prometheus.py:11-39 (ORCA150SuperDense), :365 (Prometheus)
    return {
        "sensor_pos_x": lambda x: x / 100,
        "sensor_pos_y": lambda x: x / 100,
        "sensor_pos_z": lambda x: (x + 350) / 100,
        "t": lambda x: x / 1.05e04,
    }


FEATURE_MAPS = {"IceCube86": icecube86, "IceCubeDeepCore": icecube_deepcore, "IceCubeUpgrade": icecube_upgrade,
                "ORCA150SuperDense": prometheus, "Prometheus": prometheus}


def standardize(detector: str, input_features: Tensor, input_feature_names: List[str]) -> Tensor:
    """``Detector._standardize`` (detector.py:64-77) on a COPY of ``input_features`` (fp32 ``[n, F]``)."""
    fmap = FEATURE_MAPS[detector]()
    out = input_features.to(torch.float32).clone()
    for idx, feature in enumerate(input_feature_names):
        out[:, idx] = fmap[feature](out[:, idx])          # KeyError for an unknown feature, as in the reference
    return out
