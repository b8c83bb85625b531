"""KITTI raw-drive camera ("next" row f4).

Same constructor arguments, accessors and streams as
/root/reference/src/python/pipeline/camera/kitti_single_view_camera.py:14-73: frames of
`<drive>/image_02/data` (left) and `<drive>/image_03/data` (right) in sorted order, zero-padded
from 375x1242 to the fixed 384x1280 working size (left 19, top 5, right 19, bottom 4), disparity
range 0..64, focal length / baseline from `<drive>/../calib_cam_to_cam.txt`, and ground-truth
disparity from the velodyne scan of the frame (baseline * focal / depth, 0 where there is no
return).  PNGs are decoded with Pillow (the reference uses torchvision.io.read_image, which is
not installed here; both yield the same RGB bytes), and `drive_dir` may be absolute -- the
reference resolves it against its own source tree (helpers/paths.py:9-10).
"""
from __future__ import annotations

import os
from typing import Iterator, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from helpers.kitti_calibration import focal_length_and_baseline, velodyne_depth_map
from pipeline.camera.camera import EvaluationCamera

KITTI_RAW_SHAPE = (375, 1242)            # what the ground-truth projection assumes (.py:60)
KITTI_PAD = (19, 19, 5, 4)               # F.pad order: left, right, top, bottom  (.py:23: [19, 5, 19, 4])


def list_drive_stereo_pairs(drive_dir: str) -> Tuple[List[str], List[str]]:
    """helpers/imageio_helpers.py:34-45: both image folders must exist."""
    left_dir = os.path.join(drive_dir, "image_02", "data")
    right_dir = os.path.join(drive_dir, "image_03", "data")
    if not os.path.exists(left_dir):
        raise RuntimeError(f"Folder for left images not found: {left_dir}.")
    if not os.path.exists(right_dir):
        raise RuntimeError(f"Folder for right images not found: {right_dir}.")
    return (sorted(os.path.join(left_dir, f) for f in os.listdir(left_dir)),
            sorted(os.path.join(right_dir, f) for f in os.listdir(right_dir)))


def _pad(t: torch.Tensor) -> torch.Tensor:
    return F.pad(t, KITTI_PAD, mode="constant", value=0)


def _read_png_chw_u8(path: str) -> torch.Tensor:
    from PIL import Image
    with Image.open(path) as im:
        arr = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))


class KittiSingleViewCamera(EvaluationCamera):

    def __init__(self, drive_dir: str, return_right_view: bool = False, only_one: bool = False):
        drive_dir = os.path.abspath(drive_dir)
        self._calib_dir = os.path.dirname(drive_dir.rstrip(os.sep))
        self._left_images, self._right_images = list_drive_stereo_pairs(drive_dir)
        self._return_right_view = return_right_view
        self._only_one = only_one
        self._focal_length, self._baseline = focal_length_and_baseline(self._calib_dir)

    def focal_length(self) -> float:
        return self._focal_length

    def baseline(self) -> float:
        return self._baseline

    def get_image_shape(self) -> Tuple[int, int]:
        return KITTI_RAW_SHAPE[0] + KITTI_PAD[2] + KITTI_PAD[3], KITTI_RAW_SHAPE[1] + KITTI_PAD[0] + KITTI_PAD[1]

    def get_disparity_boundaries(self) -> Tuple[int, int]:
        return 0, 64

    def stream_image_pairs(self) -> Iterator[Tuple[torch.Tensor, Optional[torch.Tensor]]]:
        for left_path, right_path in zip(self._left_images, self._right_images):
            yield self._load_view(left_path), (self._load_view(right_path) if self._return_right_view else None)
            if self._only_one:
                break

    def stream_image_pairs_with_gt_disparity(self) -> Iterator[Tuple[torch.Tensor, Optional[torch.Tensor], torch.Tensor]]:
        for left_path, right_path in zip(self._left_images, self._right_images):
            left = self._load_view(left_path)
            right = self._load_view(right_path) if self._return_right_view else None
            yield left, right, self._load_velodyne_gt_disparity_map(left_path)
            if self._only_one:
                break

    def _load_view(self, path: str) -> torch.Tensor:
        return _pad(_read_png_chw_u8(path))

    def _load_velodyne_gt_disparity_map(self, left_image_path: str) -> torch.Tensor:
        depth = torch.from_numpy(velodyne_depth_map(self._calib_dir, self._velodyne_path(left_image_path),
                                                    KITTI_RAW_SHAPE, vel_depth=True))
        disparity = self._baseline * self._focal_length / depth        # .py:68-69; depth 0 -> inf
        disparity[torch.isinf(disparity)] = 0
        return _pad(disparity)

    @staticmethod
    def _velodyne_path(left_image_path: str) -> str:
        return left_image_path.replace("image_02", "velodyne_points").replace(".png", ".bin")
