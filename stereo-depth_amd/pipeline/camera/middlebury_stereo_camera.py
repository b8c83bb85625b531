"""Middlebury single-pair camera ("next" row f4).

Reads the directory layout of the Middlebury 2014 stereo sets (im0.png, im1.png, calib.txt) and
exposes the same accessors as the reference's reader
(/root/reference/src/python/pipeline/camera/middlebury_stereo_camera.py:11-102): focal length =
fx of cam0, baseline, (height, width) and the (vmin, vmax) disparity bounds of calib.txt.  PNGs
are decoded with Pillow into uint8 [3, H, W] tensors (the reference uses
torchvision.io.read_image, which is not installed here; both yield the same RGB bytes).
"""
from __future__ import annotations

import dataclasses
import os
from typing import Callable, Dict, Iterator, Optional, Tuple

import numpy as np
import torch

from pipeline.camera.camera import Camera


def _parse_matrix(text: str) -> np.ndarray:
    """`[a b c; d e f; g h i]` -> 3x3 float array (rows separated by ';')."""
    body = text.strip().lstrip("[").rstrip("]")
    return np.array([[float(tok) for tok in row.split()] for row in body.split(";")])


@dataclasses.dataclass
class MiddleBuryStereoCameraCalibration:
    """The nine keys of a Middlebury calib.txt (reference :11-44 exposes the same attributes)."""
    cam0: np.ndarray
    cam1: np.ndarray
    doffs: float
    baseline: float
    width: int
    height: int
    ndisp: int
    vmin: int
    vmax: int


    @classmethod
    def from_file(cls, path: str) -> "MiddleBuryStereoCameraCalibration":
        parsers: Dict[str, Callable[[str], object]] = {"cam0": _parse_matrix, "cam1": _parse_matrix,
                                                       "doffs": float, "baseline": float}
        for name in ("width", "height", "ndisp", "vmin", "vmax"):
            parsers[name] = int
        values = {}
        with open(path, "r") as f:
            for raw in f:
                if not raw.strip():
                    continue
                key, _, text = raw.partition("=")
                key = key.strip()
                if key not in parsers:
                    raise KeyError(key)          # the reference's parser table fails the same way
                values[key] = parsers[key](text.strip())
        return cls(**values)

    @property
    def fx(self) -> float:
        return float(self.cam0[0, 0])

    @property
    def fy(self) -> float:
        return float(self.cam0[1, 1])

    @property
    def cx(self) -> float:
        return float(self.cam0[0, 2])

    @property
    def cy(self) -> float:
        return float(self.cam0[1, 2])

    def get_focal_length(self) -> Tuple[float, float]:
        return self.fx, self.fy

    def get_principal_point(self) -> Tuple[float, float]:
        return self.cx, self.cy


def _decode_png(path: str) -> torch.Tensor:
    from PIL import Image
    with Image.open(path) as im:
        hwc = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return torch.from_numpy(np.ascontiguousarray(hwc.transpose(2, 0, 1)))


class MiddleBuryStereoCamera(Camera):
    """One rectified pair; `stream_image_pairs` yields it once."""

    def __init__(self, middlebury_dir: str):
        if not os.path.isdir(middlebury_dir):
            raise RuntimeError(f"Directory '{middlebury_dir}' not found.")
        self._calibration = MiddleBuryStereoCameraCalibration.from_file(os.path.join(middlebury_dir, "calib.txt"))
        self._left_image = _decode_png(os.path.join(middlebury_dir, "im0.png"))
        self._right_image = _decode_png(os.path.join(middlebury_dir, "im1.png"))

    def focal_length(self) -> float:
        return self._calibration.fx

    def baseline(self) -> float:
        return self._calibration.baseline

    def get_image_shape(self) -> Tuple[int, int]:
        return self._calibration.height, self._calibration.width

    def get_disparity_boundaries(self) -> Tuple[int, int]:
        return self._calibration.vmin, self._calibration.vmax

    def stream_image_pairs(self) -> Iterator[Tuple[torch.Tensor, Optional[torch.Tensor]]]:
        yield self._left_image, self._right_image
