"""Per-detector feature standardisation (host side of row a1 of SURVEY.md §8).

Mirrors ``models/detector/detector.py:64-77`` (``Detector._standardize``: in-place,
column by column, ``KeyError`` for an unknown feature name) and the scaling constants of
``models/detector/icecube.py:11-48,116-170`` and ``models/detector/prometheus.py:11-39,365``.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Sequence, Tuple

import torch
from torch import Tensor

from .model import Model


# A standardisation is a short program of (op, constant) steps applied left to right in fp32 - the same
# expression, operation by operation, as the reference's lambdas (``x / 500.0`` is a division, never a
# multiplication by the reciprocal: the standardised coordinates decide the k-NN graph bit for bit).
Op = Tuple[str, float]
_OPS = {"add": 0, "sub": 1, "mul": 2, "div": 3, "log10": 4}


def _apply_ops(x: Tensor, prog: Sequence[Op]) -> Tensor:
    for op, c in prog:
        if op == "add":
            x = x + c
        elif op == "sub":
            x = x - c
        elif op == "mul":
            x = x * c
        elif op == "div":
            x = x / c
        elif op == "log10":
            x = torch.log10(x)
        else:
            raise ValueError(op)
    return x


class Detector(Model):
    """Base class: ``detector(input_features, input_feature_names) -> standardized tensor`` (a ``Model``, as in
    the reference, so that it appears in ``ModelConfig`` files by class name)."""

    xyz: List[str] = []
    string_id_column = "string"
    sensor_id_column = "sensor_id"

    def feature_ops(self) -> Dict[str, List[Op]]:
        """feature name -> program (``[]`` = identity)."""
        raise NotImplementedError

    def feature_map(self) -> Dict[str, Callable[[Tensor], Tensor]]:
        return {k: (lambda x, p=tuple(v): _apply_ops(x, p)) for k, v in self.feature_ops().items()}

    def forward(self, input_features: Tensor, input_feature_names: List[str]) -> Tensor:
        return self._standardize(input_features, input_feature_names)

    def _standardize(self, input_features: Tensor, input_feature_names: List[str]) -> Tensor:
        """In place, column by column (``detector.py:64-77``).  A HIP tensor is standardised by one kernel
        (``gn_standardize``) running the same programs; a CPU tensor with torch ops."""
        fops = self.feature_ops()
        for feature in input_feature_names:
            if feature not in fops:
                raise KeyError(f"No Standardization function found for '{feature}'")
        if input_features.is_cuda and input_features.dtype == torch.float32 and input_features.dim() == 2:
            from . import ops
            ops.standardize(input_features, [fops[f] for f in input_feature_names])
            return input_features
        for idx, feature in enumerate(input_feature_names):
            input_features[:, idx] = _apply_ops(input_features[:, idx], fops[feature])
        return input_features


class IceCube86(Detector):
    xyz = ["dom_x", "dom_y", "dom_z"]

    def feature_ops(self):
        xyz = [("div", 500.0)]
        return {
            "dom_x": xyz, "dom_y": xyz, "dom_z": xyz,
            "dom_time": [("sub", 1.0e04), ("div", 3.0e4)], "charge": [("log10", 0.0)],
            "rde": [("sub", 1.25), ("div", 0.25)], "pmt_area": [("div", 0.05)], "hlc": [],
        }


class IceCubeDeepCore(IceCube86):
    def feature_ops(self):
        return {
            "dom_x": [("div", 100.0)], "dom_y": [("div", 100.0)], "dom_z": [("add", 350.0), ("div", 100.0)],
            "dom_time": [("div", 1.05e04), ("sub", 1.0), ("mul", 20.0)], "charge": [],
            "rde": [("sub", 1.25), ("div", 0.25)], "pmt_area": [("div", 0.05)], "hlc": [],
        }


class IceCubeUpgrade(Detector):
    xyz = ["dom_x", "dom_y", "dom_z"]

    def feature_ops(self):
        return {
            "dom_x": [("div", 500.0)], "dom_y": [("div", 500.0)], "dom_z": [("div", 500.0)],
            "dom_time": [("div", 2e04), ("sub", 1.0)],
            "charge": [("log10", 0.0), ("div", 2.0)],
            "rde": [],
            "pmt_area": [("div", 0.05)],
            "string": [("sub", 50.0), ("div", 50.0)],
            "pmt_number": [("div", 20.0)],
            "dom_number": [("sub", 60.0), ("div", 60.0)],
            "pmt_dir_x": [], "pmt_dir_y": [], "pmt_dir_z": [],
            "dom_type": [("div", 130.0)],
            "hlc": [],
        }


class ORCA150SuperDense(Detector):
    xyz = ["sensor_pos_x", "sensor_pos_y", "sensor_pos_z"]
    string_id_column = "sensor_string_id"

    def feature_ops(self):
        return {
            "sensor_pos_x": [("div", 100)], "sensor_pos_y": [("div", 100)],
            "sensor_pos_z": [("add", 350), ("div", 100)], "t": [("div", 1.05e04)],
        }


class Prometheus(ORCA150SuperDense):
    """Reference to ORCA150SuperDense (``prometheus.py:365``)."""
