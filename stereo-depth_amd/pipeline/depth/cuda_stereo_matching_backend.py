"""The 'cuda' stereo-matching backend: mirrors
/root/reference/src/python/pipeline/depth/cuda_stereo_matching_backend.py:7-17.
`cuda_depth` here is the ctypes module over libstereo_mi355x.so; on torch-ROCm
`.cuda()` moves the tensor to the MI355X."""
from typing import Optional

import torch
import cuda_depth

from pipeline.depth.stereo_matching import StereoMatching


class CudaStereoMatchingBackend(StereoMatching):

    def __init__(self, configuration: Optional[cuda_depth.StereoMatchingConfiguration] = None):
        if configuration is None:
            configuration = cuda_depth.StereoMatchingConfiguration()
        self._stereo_algo = cuda_depth.StereoMatching(configuration)

    def process(self, left_image: torch.Tensor, right_image: torch.Tensor) -> torch.Tensor:
        left_gpu = left_image.cuda().float().contiguous()
        right_gpu = right_image.cuda().float().contiguous()
        output_disparity = self._stereo_algo.compute_disparity_map(left_gpu, right_gpu)
        return output_disparity
