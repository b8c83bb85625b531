"""The 'cuda' stereo-matching backend of the pipeline, over the MI355X engine.

Plays the role of /root/reference/src/python/pipeline/depth/cuda_stereo_matching_backend.py:7-17:
owns one native engine (here `cuda_depth` = the ctypes module over libstereo_mi355x.so) and feeds
it device-resident CHW frames.  The reference converts every frame to float32 before the call;
uint8 frames (what the cameras deliver) are handed over as they are -- the engine's RGB-u8 entry
does the same arithmetic on the bytes (float(u8) is exact), without the 4x larger copy.
"""
from __future__ import annotations

from typing import Optional

import torch

import cuda_depth
from pipeline.depth.stereo_matching import StereoMatching


def _device_frame(image: torch.Tensor) -> torch.Tensor:
    """Contiguous, on the GPU, uint8 kept, everything else as float32."""
    image = image.cuda()
    if image.dtype != torch.uint8:
        image = image.float()
    return image.contiguous()


class CudaStereoMatchingBackend(StereoMatching):

    def __init__(self, configuration: Optional["cuda_depth.StereoMatchingConfiguration"] = None):
        self._stereo_algo = cuda_depth.StereoMatching(configuration or cuda_depth.StereoMatchingConfiguration())

    def process(self, left_image: torch.Tensor, right_image: torch.Tensor) -> torch.Tensor:
        left, right = _device_frame(left_image), _device_frame(right_image)
        if left.dtype != right.dtype:                       # mixed inputs: fall back to float for both
            left, right = left.float(), right.float()
        return self._stereo_algo.compute_disparity_map(left, right)
