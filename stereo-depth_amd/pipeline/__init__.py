"""Host side of the hot path: the pipeline facade that drives the stereo-matching backend.

Public names are the ones the reference package exports
(/root/reference/src/python/pipeline/__init__.py:1), plus the result / context records.
"""
from pipeline.depth_estimation_pipeline import (
    DepthEstimationPipeline,
    DepthEstimationPipelineConfig,
    DepthEstimationPipelineContext,
    DepthEstimationResult,
)

__all__ = [
    "DepthEstimationPipeline",
    "DepthEstimationPipelineConfig",
    "DepthEstimationPipelineContext",
    "DepthEstimationResult",
]
