"""Camera interfaces: mirror of /root/reference/src/python/pipeline/camera/camera.py:7-34."""
from abc import ABC, abstractmethod
from typing import Tuple, Iterator, Optional

import torch


class Camera(ABC):

    @abstractmethod
    def focal_length(self) -> float:
        pass

    @abstractmethod
    def baseline(self) -> float:
        pass

    @abstractmethod
    def get_image_shape(self) -> Tuple[int, int]:
        pass

    @abstractmethod
    def get_disparity_boundaries(self) -> Tuple[int, int]:
        pass

    @abstractmethod
    def stream_image_pairs(self) -> Iterator[Tuple[torch.Tensor, Optional[torch.Tensor]]]:
        pass


class EvaluationCamera(Camera):

    @abstractmethod
    def stream_image_pairs_with_gt_disparity(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
        pass
