"""Disparity -> depth -> point list on the device ("next" row f3): what the reference's
PointCloudSaver computes before handing the points to Open3D
(/root/reference/src/python/pipeline/depth_estimation_pipeline_hooks.py:84-92,
helpers/point_cloud_helpers.py:5-13).  Writing .ply files (Open3D) is out of scope."""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import torch

from cuda_depth._native import LIB, check


def disparity_to_depth_and_points(disparity_map: torch.Tensor, focal_length: float, baseline: float,
                                  invalid_disparity: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns (depth [H,W] float32, points [N,3] float32 = [y, x, depth] of the valid pixels, row-major)."""
    if not disparity_map.is_cuda:
        raise RuntimeError("disparity_map must be a CUDA tensor")
    disp = disparity_map.float().contiguous()
    H, W = disp.shape
    dev = disp.device
    depth = torch.empty_like(disp)
    points = torch.empty((H * W, 3), dtype=torch.float32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    work = torch.empty(2 * H, dtype=torch.int32, device=dev)
    bf = torch.tensor(baseline * focal_length, dtype=torch.float32).item()    # python scalar -> float32, as torch does
    check(LIB.smx_disparity_to_points(dev.index, disp.data_ptr(), H, W, bf, float(invalid_disparity),
                                      depth.data_ptr(), points.data_ptr(), count.data_ptr(), work.data_ptr(),
                                      C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    n = int(count.item())
    return depth, points[:n]
