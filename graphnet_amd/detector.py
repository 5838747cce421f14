"""Per-detector feature standardisation (host side of row a1 of SURVEY.md §8).

Mirrors ``models/detector/detector.py:64-77`` (``Detector._standardize``: in-place,
column by column, ``KeyError`` for an unknown feature name) and the scaling constants of
``models/detector/icecube.py:11-48,116-170`` and ``models/detector/prometheus.py:11-39,365``.
"""
from __future__ import annotations

from typing import Callable, Dict, List

import torch
from torch import Tensor


class Detector:
    """Base class: ``detector(input_features, input_feature_names) -> standardized tensor``."""

    xyz: List[str] = []
    string_id_column = "string"
    sensor_id_column = "sensor_id"

    def feature_map(self) -> Dict[str, Callable[[Tensor], Tensor]]:
        raise NotImplementedError

    def __call__(self, input_features: Tensor, input_feature_names: List[str]) -> Tensor:
        return self._standardize(input_features, input_feature_names)

    forward = __call__

    def _standardize(self, input_features: Tensor, input_feature_names: List[str]) -> Tensor:
        fmap = self.feature_map()
        for idx, feature in enumerate(input_feature_names):
            if feature not in fmap:
                raise KeyError(f"No Standardization function found for '{feature}'")
            input_features[:, idx] = fmap[feature](input_features[:, idx])
        return input_features

    @staticmethod
    def _identity(x: Tensor) -> Tensor:
        return x


class IceCube86(Detector):
    xyz = ["dom_x", "dom_y", "dom_z"]

    def feature_map(self):
        return {
            "dom_x": self._dom_xyz, "dom_y": self._dom_xyz, "dom_z": self._dom_xyz,
            "dom_time": self._dom_time, "charge": self._charge, "rde": self._rde,
            "pmt_area": self._pmt_area, "hlc": self._identity,
        }

    @staticmethod
    def _dom_xyz(x):
        return x / 500.0

    @staticmethod
    def _dom_time(x):
        return (x - 1.0e04) / 3.0e4

    @staticmethod
    def _charge(x):
        return torch.log10(x)

    @staticmethod
    def _rde(x):
        return (x - 1.25) / 0.25

    @staticmethod
    def _pmt_area(x):
        return x / 0.05


class IceCubeDeepCore(IceCube86):
    def feature_map(self):
        return {
            "dom_x": self._dom_xy, "dom_y": self._dom_xy, "dom_z": self._dom_z,
            "dom_time": self._dom_time_dc, "charge": self._identity, "rde": self._rde,
            "pmt_area": self._pmt_area, "hlc": self._identity,
        }

    @staticmethod
    def _dom_xy(x):
        return x / 100.0

    @staticmethod
    def _dom_z(x):
        return (x + 350.0) / 100.0

    @staticmethod
    def _dom_time_dc(x):
        return ((x / 1.05e04) - 1.0) * 20.0


class IceCubeUpgrade(Detector):
    xyz = ["dom_x", "dom_y", "dom_z"]

    def feature_map(self):
        return {
            "dom_x": lambda x: x / 500.0, "dom_y": lambda x: x / 500.0, "dom_z": lambda x: x / 500.0,
            "dom_time": lambda x: (x / 2e04) - 1.0,
            "charge": lambda x: torch.log10(x) / 2.0,
            "rde": self._identity,
            "pmt_area": lambda x: x / 0.05,
            "string": lambda x: (x - 50.0) / 50.0,
            "pmt_number": lambda x: x / 20.0,
            "dom_number": lambda x: (x - 60.0) / 60.0,
            "pmt_dir_x": self._identity, "pmt_dir_y": self._identity, "pmt_dir_z": self._identity,
            "dom_type": lambda x: x / 130.0,
            "hlc": self._identity,
        }


class ORCA150SuperDense(Detector):
    xyz = ["sensor_pos_x", "sensor_pos_y", "sensor_pos_z"]
    string_id_column = "sensor_string_id"

    def feature_map(self):
        return {
            "sensor_pos_x": lambda x: x / 100, "sensor_pos_y": lambda x: x / 100,
            "sensor_pos_z": lambda x: (x + 350) / 100, "t": lambda x: x / 1.05e04,
        }


class Prometheus(ORCA150SuperDense):
    """Reference to ORCA150SuperDense (``prometheus.py:365``)."""
