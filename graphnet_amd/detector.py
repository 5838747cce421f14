"""Per-detector feature standardisation (host side of row a1 of SURVEY.md §8).

Mirrors ``models/detector/detector.py:64-77`` (``Detector._standardize``: in-place,
column by column, ``KeyError`` for an unknown feature name) and the scaling constants of
``models/detector/icecube.py:11-48,116-170`` and ``models/detector/prometheus.py:11-39,365``.
"""
from __future__ import annotations

import os
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

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
    # parquet with the xyz / string / sensor-id columns (detector.py:33-46): (experiment directory, file name) below
    # the directory named by $GRAPHNET_AMD_GEOMETRY_TABLES (the reference's data/geometry_tables), or an explicit path
    geometry_table_file: Optional[Tuple[str, str]] = None

    @property
    def geometry_table_path(self) -> Optional[str]:
        explicit = getattr(self, "_geometry_table_path", None)
        if explicit:
            return explicit
        root = os.environ.get("GRAPHNET_AMD_GEOMETRY_TABLES")
        if root and self.geometry_table_file:
            return os.path.join(root, *self.geometry_table_file)
        return None

    @geometry_table_path.setter
    def geometry_table_path(self, path: Optional[str]) -> None:
        self._geometry_table_path = path
        self._geometry_table = None

    @property
    def geometry_table(self):
        """Sensor table (pandas), read once from ``geometry_table_path``; may also be assigned
        (``detector.geometry_table = frame``) where the reference's data directory is not installed."""
        if getattr(self, "_geometry_table", None) is None:
            path = self.geometry_table_path
            if not path:
                raise AttributeError(f"{self.__class__.__name__}: no geometry table - set $GRAPHNET_AMD_GEOMETRY_TABLES, "
                                     "detector.geometry_table_path or detector.geometry_table = <DataFrame> before "
                                     "using sensor / string masks or inactive sensors")
            import pandas as pd
            self._geometry_table = pd.read_parquet(path)
        return self._geometry_table

    @geometry_table.setter
    def geometry_table(self, frame: Any) -> None:
        self._geometry_table = frame

    @property
    def sensor_position_names(self) -> List[str]:
        return list(self.xyz)

    @property
    def sensor_index_name(self) -> str:
        return self.sensor_id_column

    @property
    def string_index_name(self) -> str:
        return self.string_id_column

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
    geometry_table_file = ("icecube", "icecube86.parquet")

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
    geometry_table_file = ("icecube", "icecube_upgrade.parquet")

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
    geometry_table_file = ("prometheus", "orca_150.parquet")
    string_id_column = "sensor_string_id"

    def feature_ops(self):
        return {
            "sensor_pos_x": [("div", 100)], "sensor_pos_y": [("div", 100)],
            "sensor_pos_z": [("add", 350), ("div", 100)], "t": [("div", 1.05e04)],
        }


class Prometheus(ORCA150SuperDense):
    """Reference to ORCA150SuperDense (``prometheus.py:365``)."""
