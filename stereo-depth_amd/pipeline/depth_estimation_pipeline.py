"""Pipeline facade: the direct caller of the backend boundary.  Mirrors
/root/reference/src/python/pipeline/depth_estimation_pipeline.py:14-87 for the 'cuda'
backend.  Right-view synthesis (Deep3D) and the DNN backends are out of scope
(SURVEY.md section 2), so `right_image` is mandatory and other backend names raise."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Literal, Tuple, Optional, Any

import torch
import cuda_depth

from helpers.torch_helpers import cuda_perf_clock
from pipeline.depth import StereoMatching, CudaStereoMatchingBackend, AVAILABLE_DNN_BACKENDS


@dataclass
class DepthEstimationPipelineConfig:
    image_shape: Tuple[int, int] = (384, 1280)
    min_disparity: int = 1
    max_disparity: int = 64
    invalid_disparity: float = -1.0
    stereo_matching_backend: Literal["msnet2d", "msnet3d", "gwcnet", "cuda"] = "cuda"
    log_perf_time: bool = False

    def update(self, **kwargs: Any) -> DepthEstimationPipelineConfig:
        for (key, value) in kwargs.items():
            if not hasattr(self, key):
                raise RuntimeError(f"Unexpected keyword argument: '{key}'.")
            setattr(self, key, value)
        return self


@dataclass
class DepthEstimationResult:
    left_image: torch.Tensor
    right_image: torch.Tensor
    disparity_map: torch.Tensor


@dataclass
class DepthEstimationPipelineContext:
    disparity_map: torch.Tensor
    left_image: torch.Tensor
    right_image: torch.Tensor
    config: DepthEstimationPipelineConfig
    frame_index: int


class DepthEstimationPipeline:

    def __init__(self, config: Optional[DepthEstimationPipelineConfig] = None):
        self._config = config if config is not None else DepthEstimationPipelineConfig()
        self._stereo_matching = self._get_stereo_matching()
        print(f"Using '{self._config.stereo_matching_backend}' as stereo matching backend.")

    def process(self, left_image: torch.Tensor, right_image: Optional[torch.Tensor] = None) -> DepthEstimationResult:
        left_image = left_image.cuda()
        if right_image is None:
            raise RuntimeError("right_image is required: right-view synthesis (Deep3D) is not part of this build.")
        with cuda_perf_clock("Stereo matching", self._config.log_perf_time):
            disparity_map = self._stereo_matching.process(left_image, right_image)
        return DepthEstimationResult(
            disparity_map=disparity_map,
            left_image=left_image,
            right_image=right_image
        )

    def get_configuration(self) -> DepthEstimationPipelineConfig:
        return self._config

    def _get_stereo_matching(self) -> StereoMatching:
        if self._config.stereo_matching_backend in AVAILABLE_DNN_BACKENDS:
            raise RuntimeError(f"Stereo matching backend '{self._config.stereo_matching_backend}' "
                               f"(traced DNN) is not part of this build; use 'cuda'.")
        elif self._config.stereo_matching_backend == "cuda":
            config = cuda_depth.StereoMatchingConfiguration(
                height=self._config.image_shape[0],
                width=self._config.image_shape[1],
                min_disparity=self._config.min_disparity,
                max_disparity=self._config.max_disparity,
            )
            stereo_matching = CudaStereoMatchingBackend(configuration=config)
            return stereo_matching
        else:
            raise RuntimeError(f"Unsupported stereo matching backend: {self._config.stereo_matching_backend}")
