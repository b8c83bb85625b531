"""Plugin boundary (Python side): mirrors /root/reference/src/python/pipeline/depth/stereo_matching.py:6-10."""
from abc import ABC, abstractmethod

import torch


class StereoMatching(ABC):

    @abstractmethod
    def process(self, left_image: torch.Tensor, right_image: torch.Tensor) -> torch.Tensor:
        pass
