"""The Python side of the plugin boundary.

One abstract type with one method, `process(left, right) -> disparity`, exactly what the
reference's pipeline calls on whichever backend it was configured with
(/root/reference/src/python/pipeline/depth/stereo_matching.py:6-10).
"""
from __future__ import annotations

import abc

import torch


class StereoMatching(metaclass=abc.ABCMeta):
    """A stereo matcher: two rectified views in, one dense disparity map out."""

    @abc.abstractmethod
    def process(self, left_image: torch.Tensor, right_image: torch.Tensor) -> torch.Tensor:
        """left_image, right_image: [3, H, W] (any dtype / device the backend accepts).
        Returns the [H, W] float32 disparity map on the GPU."""

    def __call__(self, left_image: torch.Tensor, right_image: torch.Tensor) -> torch.Tensor:
        return self.process(left_image, right_image)
