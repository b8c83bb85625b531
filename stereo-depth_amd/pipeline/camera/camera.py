"""Camera interfaces of the depth-estimation pipeline.

Same two abstract types and method names as the reference
(/root/reference/src/python/pipeline/camera/camera.py:7-34), so camera implementations and the
runner (depth_estimation_pipeline_runner.py) are interchangeable with the reference's:

    Camera             focal_length(), baseline(), get_image_shape(), get_disparity_boundaries(),
                       stream_image_pairs()
    EvaluationCamera   + stream_image_pairs_with_gt_disparity()
"""
from __future__ import annotations

import abc
from typing import Iterator, Optional, Tuple

import torch

ImagePair = Tuple[torch.Tensor, Optional[torch.Tensor]]                 # (left CHW u8, right CHW u8 or None)
ImagePairWithDisparity = Tuple[torch.Tensor, Optional[torch.Tensor], torch.Tensor]


class Camera(metaclass=abc.ABCMeta):
    """A calibrated source of rectified stereo frames."""

    @abc.abstractmethod
    def focal_length(self) -> float:
        """Horizontal focal length in pixels (depth = baseline * focal_length / disparity)."""

    @abc.abstractmethod
    def baseline(self) -> float:
        """Distance between the two optical centres, in the unit depth is reported in."""

    @abc.abstractmethod
    def get_image_shape(self) -> Tuple[int, int]:
        """(height, width) of every streamed frame -- what the stereo engine is configured with."""

    @abc.abstractmethod
    def get_disparity_boundaries(self) -> Tuple[int, int]:
        """(min_disparity, max_disparity) the matcher has to search."""

    @abc.abstractmethod
    def stream_image_pairs(self) -> Iterator[ImagePair]:
        """Frames in capture order; the right view is None for single-view sources."""


class EvaluationCamera(Camera):
    """A camera that also knows the ground-truth disparity of its frames (0 = no measurement)."""

    @abc.abstractmethod
    def stream_image_pairs_with_gt_disparity(self) -> Iterator[ImagePairWithDisparity]:
        """(left, right or None, ground-truth disparity [H, W]) per frame."""
