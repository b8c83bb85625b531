"""Middlebury single-pair camera ("next" row f4): same directory layout (im0.png, im1.png,
calib.txt), calibration keys and accessors as
/root/reference/src/python/pipeline/camera/middlebury_stereo_camera.py:11-102.  PNGs are decoded
with Pillow into uint8 [3,H,W] tensors (the reference uses torchvision.io.read_image, which is not
installed here; both yield the same RGB bytes)."""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Tuple, Iterator, Optional

import numpy as np
import torch

from pipeline.camera.camera import Camera


@dataclass
class MiddleBuryStereoCameraCalibration:
    cam0: np.ndarray
    cam1: np.ndarray
    doffs: float
    baseline: float
    width: int
    height: int
    ndisp: int
    vmin: int
    vmax: int

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


def _read_image_chw_u8(path: str) -> torch.Tensor:
    from PIL import Image
    with Image.open(path) as im:
        arr = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))


class MiddleBuryStereoCamera(Camera):

    def __init__(self, middlebury_dir: str):
        if not os.path.exists(middlebury_dir):
            raise RuntimeError(f"Directory '{middlebury_dir}' not found.")
        self._left_image = _read_image_chw_u8(os.path.join(middlebury_dir, "im0.png"))
        self._right_image = _read_image_chw_u8(os.path.join(middlebury_dir, "im1.png"))
        self._calibration = MiddleBuryStereoCamera._load_calibration_file(os.path.join(middlebury_dir, "calib.txt"))

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

    @staticmethod
    def _load_camera_intrinsics(intrinsics: str) -> np.ndarray:
        return np.array(
            [[float(x.strip()) for x in arr.strip().split(" ")]
             for arr in intrinsics.replace("[", "").replace("]", "").split(";")]
        )

    @staticmethod
    def _load_calibration_file(calibration_file_path: str) -> MiddleBuryStereoCameraCalibration:
        parsers = {
            "cam0": MiddleBuryStereoCamera._load_camera_intrinsics,
            "cam1": MiddleBuryStereoCamera._load_camera_intrinsics,
            "doffs": float, "baseline": float, "width": int, "height": int,
            "ndisp": int, "vmin": int, "vmax": int,
        }
        data = {}
        with open(calibration_file_path, "r") as calibration_file:
            for line in calibration_file:
                if not line.strip():
                    continue
                key, value = line.split("=")
                data[key] = parsers[key](value)
        return MiddleBuryStereoCameraCalibration(**data)
