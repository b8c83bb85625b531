"""Evaluation metrics (D1, Threshold_N, MAE) -- mirrors
/root/reference/src/python/pipeline/depth_estimation_pipeline_metrics.py:7-56 (same class names,
`process(disparity_estimate, disparity_gt, mask) -> float`, `name()`), computed by one fused HIP
pass over the disparity map (`smx_eval_metrics`) instead of boolean-index gathers + reductions.
`FusedDisparityMetrics.evaluate` returns all six numbers the reference's evaluator reports
(scripts/evaluate_depth_estimation_pipeline.py) from a single kernel launch.
"""
from __future__ import annotations

import ctypes as C
from abc import ABC, abstractmethod
from typing import Dict, Optional, Sequence

import torch

from cuda_depth._native import LIB, check


class FusedDisparityMetrics:
    """counts/sums for one or more images: D1, four thresholds, L1."""

    @staticmethod
    def sums(disparity_estimate: torch.Tensor, disparity_gt: torch.Tensor, mask: Optional[torch.Tensor] = None,
             max_disparity: float = float("inf"), thresholds: Sequence[float] = (1.0, 2.0, 3.0, 5.0)) -> torch.Tensor:
        est = disparity_estimate.float().contiguous()
        gt = disparity_gt.float().contiguous()
        if not est.is_cuda or not gt.is_cuda:
            raise RuntimeError("metrics inputs must be CUDA tensors")
        if est.shape != gt.shape:
            raise RuntimeError(f"shape mismatch: {tuple(est.shape)} vs {tuple(gt.shape)}")
        n = est.shape[0] if est.dim() == 3 else 1
        pixels = est.numel() // n
        mptr = None
        if mask is not None:
            if mask.shape != est.shape:
                raise RuntimeError("mask shape mismatch")
            mask = mask.to(torch.bool).contiguous()
            mptr = mask.data_ptr()
        out = torch.empty((n, 8), dtype=torch.float64, device=est.device)
        thr = (C.c_float * 4)(*[float(t) for t in thresholds])
        check(LIB.smx_eval_metrics(est.device.index, n, est.data_ptr(), gt.data_ptr(), mptr, pixels,
                                   float(max_disparity), thr, out.data_ptr(),
                                   C.c_void_p(torch.cuda.current_stream(est.device).cuda_stream)))
        return out

    @staticmethod
    def evaluate(disparity_estimate: torch.Tensor, disparity_gt: torch.Tensor, max_disparity: float) -> Dict[str, float]:
        """runner.py:82-94 for one frame: gt_mask = (gt <= max_disp) & (gt > 0), all six metrics."""
        s = FusedDisparityMetrics.sums(disparity_estimate, disparity_gt, None, max_disparity).sum(dim=0).cpu()
        cnt = float(s[0])
        f32 = lambda v: float(torch.tensor(v, dtype=torch.float32) / torch.tensor(cnt, dtype=torch.float32))
        return {"D1": f32(float(s[1])), "Threshold_1": f32(float(s[2])), "Threshold_2": f32(float(s[3])),
                "Threshold_3": f32(float(s[4])), "Threshold_5": f32(float(s[5])), "MAE": float(s[6]) / cnt if cnt else float("nan")}


class DepthEstimationPipelineMetric(ABC):

    @abstractmethod
    def process(self, disparity_estimate: torch.Tensor, disparity_gt: torch.Tensor, mask: torch.Tensor) -> float:
        pass

    @abstractmethod
    def name(self) -> str:
        pass


def _ratio(count: float, total: float) -> float:
    # torch.mean(err_mask.float()) of the reference: float32 sum of 0/1 values (exact) / float32 N
    return float(torch.tensor(count, dtype=torch.float32) / torch.tensor(total, dtype=torch.float32))


class D1Metric(DepthEstimationPipelineMetric):

    def process(self, disparity_estimate: torch.Tensor, disparity_gt: torch.Tensor, mask: torch.Tensor) -> float:
        s = FusedDisparityMetrics.sums(disparity_estimate, disparity_gt, mask).sum(dim=0).cpu()
        return _ratio(float(s[1]), float(s[0]))

    def name(self) -> str:
        return "D1"


class ThresholdMetric(DepthEstimationPipelineMetric):

    def __init__(self, threshold: float):
        super(ThresholdMetric, self).__init__()
        self._threshold = threshold

    def process(self, disparity_estimate: torch.Tensor, disparity_gt: torch.Tensor, mask: torch.Tensor) -> float:
        s = FusedDisparityMetrics.sums(disparity_estimate, disparity_gt, mask,
                                       thresholds=(self._threshold, 0.0, 0.0, 0.0)).sum(dim=0).cpu()
        return _ratio(float(s[2]), float(s[0]))

    def name(self) -> str:
        return f"Threshold_{int(self._threshold)}"


class MAEMetric(DepthEstimationPipelineMetric):

    def process(self, disparity_estimate: torch.Tensor, disparity_gt: torch.Tensor, mask: torch.Tensor) -> float:
        s = FusedDisparityMetrics.sums(disparity_estimate, disparity_gt, mask).sum(dim=0).cpu()
        # F.l1_loss over an empty selection is NaN in the reference (mean of nothing), not an exception
        return float(s[6]) / float(s[0]) if float(s[0]) else float("nan")

    def name(self) -> str:
        return "MAE"
