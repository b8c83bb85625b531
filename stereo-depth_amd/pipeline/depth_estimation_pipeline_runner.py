"""Frame loop and evaluation loop over a camera.

Entry points and their behaviour follow
/root/reference/src/python/pipeline/depth_estimation_pipeline_runner.py:12-94
(`extract_config_from_camera`, `validate_pipeline_config_wrt_camera`, `reduce_metrics`,
`run_depth_estimation_pipeline`, `run_depth_estimation_pipeline_evaluation`), without the hook
fan-out (result savers / joblib are out of scope): the frame loop returns the
per-frame results instead.  Metrics are the fused device metrics of
pipeline.depth_estimation_pipeline_metrics.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional

from pipeline import DepthEstimationPipeline, DepthEstimationPipelineConfig
from pipeline.camera.camera import Camera, EvaluationCamera
from pipeline.depth_estimation_pipeline import DepthEstimationResult
from pipeline.depth_estimation_pipeline_metrics import DepthEstimationPipelineMetric


def extract_config_from_camera(camera: Camera) -> DepthEstimationPipelineConfig:
    """A pipeline configuration for this camera's frame size and disparity range (reference :12-19)."""
    lo, hi = camera.get_disparity_boundaries()
    return DepthEstimationPipelineConfig(image_shape=camera.get_image_shape(), min_disparity=lo, max_disparity=hi)


def validate_pipeline_config_wrt_camera(config: DepthEstimationPipelineConfig, camera: Camera) -> None:
    """The engine is built for one frame size (reference :22-25)."""
    provided = camera.get_image_shape()
    if provided != config.image_shape:
        raise RuntimeError("Incompatible image shapes between pipeline configuration and camera."
                           f"Pipeline expects: {config.image_shape} but camera provides: {provided}.")


def reduce_metrics(metrics_results: Dict[str, List[float]], reduction: str) -> Dict[str, float]:
    """Per-metric "mean" or "sum" over the frames (reference :28-35); other names raise KeyError."""
    if reduction == "mean":
        return {name: sum(values) / len(values) for name, values in metrics_results.items()}
    if reduction == "sum":
        return {name: sum(values) for name, values in metrics_results.items()}
    raise KeyError(reduction)


def run_depth_estimation_pipeline(camera: Camera, pipeline: DepthEstimationPipeline) -> List[DepthEstimationResult]:
    """Every frame of the camera through the pipeline (reference :38-66 minus hooks).  Disparity
    maps are cloned: the backend returns its persistent output buffer."""
    validate_pipeline_config_wrt_camera(pipeline.get_configuration(), camera)
    collected: List[DepthEstimationResult] = []
    for left_view, right_view in camera.stream_image_pairs():
        frame = pipeline.process(left_view, right_view)
        collected.append(DepthEstimationResult(left_image=frame.left_image, right_image=frame.right_image,
                                               disparity_map=frame.disparity_map.clone()))
    return collected


def run_depth_estimation_pipeline_evaluation(camera: EvaluationCamera,
                                             pipeline: DepthEstimationPipeline,
                                             metrics: Optional[Iterable[DepthEstimationPipelineMetric]] = None,
                                             reduction: str = "mean",
                                             verbose: bool = True) -> Dict[str, float]:
    """Metrics against the camera's ground truth, on pixels with 0 < gt <= max_disparity
    (reference :69-94), reduced over the frames."""
    metric_list = list(metrics) if metrics is not None else []
    config = pipeline.get_configuration()
    validate_pipeline_config_wrt_camera(config, camera)
    per_frame: Dict[str, List[float]] = {m.name(): [] for m in metric_list}
    for index, (left_view, right_view, gt) in enumerate(camera.stream_image_pairs_with_gt_disparity()):
        gt = gt.cuda()
        valid = (gt > 0) & (gt <= config.max_disparity)
        estimate = pipeline.process(left_view, right_view).disparity_map
        for m in metric_list:
            per_frame[m.name()].append(m.process(estimate, gt, valid))
        if verbose:
            print(f"Processed frame {index}.")
    return reduce_metrics(per_frame, reduction)
