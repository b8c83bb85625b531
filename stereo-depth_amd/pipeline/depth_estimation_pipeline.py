"""Pipeline facade: the direct caller of the backend boundary.

Same public surface as /root/reference/src/python/pipeline/depth_estimation_pipeline.py:14-87
(`DepthEstimationPipelineConfig` with its six fields and `update`, `DepthEstimationResult`,
`DepthEstimationPipelineContext`, `DepthEstimationPipeline.process / get_configuration`) for the
'cuda' backend.  Right-view synthesis (Deep3D) and the traced-DNN backends are out of scope
(SURVEY.md section 2): `right_image` is mandatory here and the other backend names raise.
"""
from __future__ import annotations

import dataclasses
from typing import Any, Optional, Tuple

import torch

import cuda_depth
from helpers.torch_helpers import cuda_perf_clock
from pipeline.depth import AVAILABLE_DNN_BACKENDS, CudaStereoMatchingBackend, StereoMatching

_BACKENDS = ("cuda",) + AVAILABLE_DNN_BACKENDS


@dataclasses.dataclass
class DepthEstimationPipelineConfig:
    """Field names and defaults: depth_estimation_pipeline.py:15-21 of the reference."""
    image_shape: Tuple[int, int] = (384, 1280)
    min_disparity: int = 1
    max_disparity: int = 64
    invalid_disparity: float = -1.0
    stereo_matching_backend: str = "cuda"          # one of "cuda", "msnet2d", "msnet3d", "gwcnet"
    log_perf_time: bool = False

    def update(self, **changes: Any) -> "DepthEstimationPipelineConfig":
        """In-place update that rejects unknown fields (reference :23-28); returns self."""
        known = {f.name for f in dataclasses.fields(self)}
        unknown = [name for name in changes if name not in known]
        if unknown:
            raise RuntimeError(f"Unexpected keyword argument: '{unknown[0]}'.")
        for name, value in changes.items():
            setattr(self, name, value)
        return self

    def engine_configuration(self) -> "cuda_depth.StereoMatchingConfiguration":
        """What the pipeline hands to the native engine (reference :77-82): shape and disparity
        range; every other engine parameter keeps its default."""
        height, width = self.image_shape
        return cuda_depth.StereoMatchingConfiguration(height=height, width=width,
                                                      min_disparity=self.min_disparity,
                                                      max_disparity=self.max_disparity)


@dataclasses.dataclass
class DepthEstimationResult:
    left_image: torch.Tensor
    right_image: torch.Tensor
    disparity_map: torch.Tensor


@dataclasses.dataclass
class DepthEstimationPipelineContext:
    disparity_map: torch.Tensor
    left_image: torch.Tensor
    right_image: torch.Tensor
    config: DepthEstimationPipelineConfig
    frame_index: int


def _make_backend(config: DepthEstimationPipelineConfig) -> StereoMatching:
    name = config.stereo_matching_backend
    if name == "cuda":
        return CudaStereoMatchingBackend(configuration=config.engine_configuration())
    if name in AVAILABLE_DNN_BACKENDS:
        raise RuntimeError(f"Stereo matching backend '{name}' (traced DNN) is not part of this build; use 'cuda'.")
    raise RuntimeError(f"Unsupported stereo matching backend: {name}")


class DepthEstimationPipeline:

    def __init__(self, config: Optional[DepthEstimationPipelineConfig] = None):
        self._config = DepthEstimationPipelineConfig() if config is None else config
        self._stereo_matching = _make_backend(self._config)
        print(f"Using '{self._config.stereo_matching_backend}' as stereo matching backend.")

    def get_configuration(self) -> DepthEstimationPipelineConfig:
        return self._config

    def process(self, left_image: torch.Tensor, right_image: Optional[torch.Tensor] = None) -> DepthEstimationResult:
        """One frame.  The returned disparity map aliases the engine's persistent output buffer
        (stereo_matching.cc:42): clone it before processing the next frame if it must survive."""
        if right_image is None:
            raise RuntimeError("right_image is required: right-view synthesis (Deep3D) is not part of this build.")
        left_on_device = left_image.cuda()
        with cuda_perf_clock("Stereo matching", self._config.log_perf_time):
            disparity = self._stereo_matching.process(left_on_device, right_image)
        return DepthEstimationResult(left_image=left_on_device, right_image=right_image, disparity_map=disparity)
